// pipe.hip -- the layer pipeline under the C ABI: RCCL point-to-point directly, no torch.
// New functionality (the reference is single-device, gpu.rs:215).
//
// Partition: contiguous layer ranges, as even as possible, the first L mod N ranks one layer more (pipeline.py split_layers,
// host/main.cpp): rank r owns its range with the weights and KV slabs, rank 0 the
// embedding table, the last rank the final norm + classifier.  A stage boundary is one ncclSend /
// ncclRecv of the residual x[dim] (16 KiB at llama2-7B) to the next rank, and of the sampled token
// id (4 bytes) from the last rank back to rank 0 -- point-to-point over xGMI, no collective.
// Batch-1 decode is sequential in the layers, so n_seq >= N sequences are kept in flight to keep every
// stage busy: item j = (slot j % S, position j / S) with S = max(n_seq, N) slots per round (slots
// beyond n_seq idle: a single sequence works too, N - 1 of N ticks idle per stage); rank r computes
// item tick - r at each tick, then
// posts ONE grouped exchange (its output to the next rank + the receive of its next input), which
// is deadlock-free for any N including N = 2, where both directions share one peer.  Everything is
// enqueued on the context's stream: no host synchronisation inside the loop.
//
// RCCL is loaded with dlopen on first use, so the single-GPU product has no RCCL dependency.
// (A process that also imports PyTorch should import it first: torch ships its own librccl.so.1, the
// loader serves every later request for that soname from the copy already in the process, and torch
// on the other build crashes at exit.  dlopen("librccl.so.1") here then simply reuses torch's copy.)
#include "../../include/rama_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <strings.h>

extern "C" void* rama_internal_stream(rama_ctx* c);      // rama_api.hip
extern "C" int rama_internal_device(rama_ctx* c);

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;         // optional
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;      // optional
};
Rccl g_rccl;
std::mutex g_rccl_mu;
thread_local std::string g_pipe_err;

int bad(int code, const std::string& msg) {
    g_pipe_err = msg;
    fprintf(stderr, "rama_pipe: %s\n", msg.c_str());
    return code;
}

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return 0;
    // RCCL prints a version banner on STDOUT at NCCL_DEBUG=VERSION and =WARN (some images export one of
    // them): it would land in the middle of the generated text.  RAMA_NCCL_DEBUG passes a level through.
    // The user's own NCCL_DEBUG is respected (INFO / TRACE go to stderr and are what one needs when a rendezvous hangs);
    // only the two banner levels, or no setting at all, are replaced.
    const char* dbg = getenv("RAMA_NCCL_DEBUG");
    const char* cur = getenv("NCCL_DEBUG");
    if (dbg) setenv("NCCL_DEBUG", dbg, 1);
    else if (!cur || !strcasecmp(cur, "VERSION") || !strcasecmp(cur, "WARN")) setenv("NCCL_DEBUG", "NONE", 1);
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return bad(RAMA_EIO, std::string("cannot load librccl: ") + dlerror());
    Rccl r;
    r.h = h;
#define SYM(field, name) *(void**)(&r.field) = dlsym(h, name); if (!r.field) { dlclose(h); return bad(RAMA_EIO, "librccl lacks " name); }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    *(void**)(&r.CommCount) = dlsym(h, "ncclCommCount");
    *(void**)(&r.CommUserRank) = dlsym(h, "ncclCommUserRank");
    g_rccl = r;
    return 0;
}

#define NCHK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return bad(RAMA_EIO, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); } while (0)

}  // namespace

struct rama_pipe {
    rama_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

static_assert(RAMA_PIPE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rama_hip.h must carry RCCL's unique-id size");

extern "C" int rama_pipe_unique_id(void* id_out) {
    if (!id_out) return bad(RAMA_EINVAL, "rama_pipe_unique_id: NULL argument");
    int rc = load_rccl(); if (rc) return rc;
    ncclUniqueId id;
    NCHK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return 0;
}

extern "C" int rama_pipe_create(rama_ctx* ctx, const void* id_bytes, int rank, int world, rama_pipe** out) {
    if (!ctx || !id_bytes || !out || world < 1 || rank < 0 || rank >= world) return bad(RAMA_EINVAL, "rama_pipe_create: bad argument");
    int rc = load_rccl(); if (rc) return rc;
    if (hipSetDevice(rama_internal_device(ctx)) != hipSuccess) return bad(RAMA_EIO, "rama_pipe_create: hipSetDevice failed");
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    rama_pipe* p = new rama_pipe();
    p->ctx = ctx; p->rank = rank; p->world = world;
    ncclResult_t r = g_rccl.CommInitRank(&p->comm, world, id, rank);
    if (r != ncclSuccess) { delete p; return bad(RAMA_EIO, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); }
    *out = p;
    return 0;
}

// what the communicator itself says about its size and this end's rank (the bench line's `rccl_ranks`)
extern "C" int rama_pipe_comm_info(const rama_pipe* p, int* n_ranks, int* rank) {
    if (!p || !p->comm) return bad(RAMA_EINVAL, "rama_pipe_comm_info: no communicator");
    int n = p->world, r = p->rank;
    if (g_rccl.CommCount) NCHK(g_rccl.CommCount(p->comm, &n));
    if (g_rccl.CommUserRank) NCHK(g_rccl.CommUserRank(p->comm, &r));
    if (n_ranks) *n_ranks = n;
    if (rank) *rank = r;
    return 0;
}

extern "C" int rama_pipe_destroy(rama_pipe* p) {
    if (!p) return 0;
    rama_sync(p->ctx);
    if (p->comm) g_rccl.CommDestroy(p->comm);
    delete p;
    return 0;
}

extern "C" int rama_pipe_exchange(rama_pipe* p, const float* send_x, size_t n_send_x, int send_x_peer,
                                  float* recv_x, size_t n_recv_x, int recv_x_peer,
                                  const int32_t* send_tok, int send_tok_peer, int32_t* recv_tok, int recv_tok_peer) {
    if (!p) return bad(RAMA_EINVAL, "rama_pipe_exchange: NULL pipe");
    if (!send_x && !recv_x && !send_tok && !recv_tok) return 0;
    hipStream_t st = (hipStream_t)rama_internal_stream(p->ctx);
    auto peer_ok = [&](int peer) { return peer >= 0 && peer < p->world; };
    if ((send_x && !peer_ok(send_x_peer)) || (recv_x && !peer_ok(recv_x_peer)) || (send_tok && !peer_ok(send_tok_peer)) ||
        (recv_tok && !peer_ok(recv_tok_peer)))
        return bad(RAMA_EINVAL, "rama_pipe_exchange: peer outside the communicator");
    NCHK(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    if (send_x) r = g_rccl.Send(send_x, n_send_x, ncclFloat32, send_x_peer, p->comm, st);
    if (r == ncclSuccess && send_tok) r = g_rccl.Send(send_tok, 1, ncclInt32, send_tok_peer, p->comm, st);
    if (r == ncclSuccess && recv_x) r = g_rccl.Recv(recv_x, n_recv_x, ncclFloat32, recv_x_peer, p->comm, st);
    if (r == ncclSuccess && recv_tok) r = g_rccl.Recv(recv_tok, 1, ncclInt32, recv_tok_peer, p->comm, st);
    ncclResult_t e = g_rccl.GroupEnd();
    if (r != ncclSuccess) return bad(RAMA_EIO, std::string("ncclSend/ncclRecv: ") + g_rccl.GetErrorString(r));
    if (e != ncclSuccess) return bad(RAMA_EIO, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(e));
    return 0;
}

// ---- the schedule (pure bookkeeping, mirrored by rama_amd/pipeline.py Schedule and its gloo tests)
namespace {
struct Item { bool on; int seq, pos; };
int slots_of(const rama_pipe_plan& pl, int world) { return pl.n_seq > world ? pl.n_seq : world; }
Item item_of(const rama_pipe_plan& pl, int world, int rank, int tick) {
    const long S = slots_of(pl, world), j = (long)tick - rank, total = S * pl.n_pos;
    if (j < 0 || j >= total) return {false, 0, 0};
    const int seq = (int)(j % S);
    return {seq < pl.n_seq, seq, (int)(j / S)};
}
}  // namespace

// the schedule as pure arithmetic (no GPU, no communicator): what `rank` of `world` computes at `tick`.
// Returns 1 and fills seq / pos when there is an item, 0 for an idle tick.
extern "C" int rama_pipe_item(const rama_pipe_plan* plan, int world, int rank, int tick, int* seq, int* pos) {
    if (!plan || world < 1 || rank < 0 || rank >= world || plan->n_seq < 1 || plan->n_pos < 1) return -1;
    const Item it = item_of(*plan, world, rank, tick);
    if (seq) *seq = it.seq;
    if (pos) *pos = it.pos;
    return it.on ? 1 : 0;
}

extern "C" int rama_pipe_total_ticks(const rama_pipe* p, const rama_pipe_plan* plan) {
    if (!p || !plan) return -1;
    return slots_of(*plan, p->world) * plan->n_pos + p->world - 1;
}
extern "C" int rama_pipe_plan_ticks(const rama_pipe_plan* plan, int world) {
    if (!plan || world < 1 || plan->n_seq < 1 || plan->n_pos < 1) return -1;
    return slots_of(*plan, world) * plan->n_pos + world - 1;
}

// one tick of one rank, decided: what it computes, with which token, whether it samples, what it sends and what it receives.  The ONLY place the
// native loop's decisions are made -- rama_pipe_run_ticks below executes this, tests/test_pipeline_gloo.py lets gloo ranks execute it on the CPU.
extern "C" int rama_pipe_tick_plan(const rama_pipe_plan* plan, int world, int rank, int tick, rama_pipe_tick* out) {
    if (!plan || !out || world < 1 || rank < 0 || rank >= world || plan->n_seq < 1 || plan->n_pos < 1 || plan->n_prompt < 0 || (plan->n_prompt && !plan->prompt)) return -1;
    const int last = world - 1;
    rama_pipe_tick t{};
    const Item it = item_of(*plan, world, rank, tick);
    t.on = it.on ? 1 : 0; t.seq = it.seq; t.pos = it.pos; t.token_kind = 2;
    if (it.on) {
        t.pos_wrapped = plan->wrap > 0 ? it.pos % plan->wrap : it.pos;
        if (rank == 0 && t.pos_wrapped == 0) { t.token_kind = 0; t.token = 1; }                                                        // BOS, mod.rs:182
        else if (rank == 0 && t.pos_wrapped <= plan->n_prompt) { t.token_kind = 1; t.token = plan->prompt[t.pos_wrapped - 1]; }       // mod.rs:190-191
        t.samples = rank == last ? 1 : 0;
        if (world > 1) {      // what I send after computing
            if (rank < last) { t.send_kind = RAMA_PIPE_X; t.send_seq = it.seq; t.send_peer = rank + 1; }
            else { t.send_kind = RAMA_PIPE_TOKEN; t.send_seq = it.seq; t.send_peer = 0; }
        }
    }
    if (world > 1) {          // what my upstream neighbour sends me this tick
        const int src = rank > 0 ? rank - 1 : last;
        const Item up = item_of(*plan, world, src, tick);
        if (up.on) { t.recv_kind = rank > 0 ? RAMA_PIPE_X : RAMA_PIPE_TOKEN; t.recv_seq = up.seq; t.recv_peer = src; t.recv_pos = up.pos; }
    }
    *out = t;
    return 0;
}

extern "C" int rama_pipe_run_ticks(rama_pipe* p, const rama_config* cfg, const rama_weights* w, rama_run_state* states,
                                   int32_t* const* tok_dev, const rama_stage* stage, const rama_pipe_plan* plan,
                                   int tick_from, int tick_to) {
    if (!p || !cfg || !w || !states || !tok_dev || !stage || !plan) return bad(RAMA_EINVAL, "rama_pipe_run_ticks: NULL argument");
    if (plan->n_seq < 1 || plan->n_pos < 1) return bad(RAMA_EINVAL, "rama_pipe_run_ticks: n_seq and n_pos must be positive");
    if (plan->n_prompt < 0 || (plan->n_prompt && !plan->prompt)) return bad(RAMA_EINVAL, "rama_pipe_run_ticks: bad prompt");
    const int rank = p->rank, world = p->world;
    hipStream_t stream = (hipStream_t)rama_internal_stream(p->ctx);
    for (int tick = tick_from; tick < tick_to; tick++) {
        rama_pipe_tick t;
        if (rama_pipe_tick_plan(plan, world, rank, tick, &t)) return bad(RAMA_EINVAL, "rama_pipe_run_ticks: bad plan");
        if (t.on) {
            rama_run_state* st = &states[t.seq];
            int rc;
            if (t.token_kind != 2) rc = rama_forward_stage(p->ctx, cfg, w, st, t.token, t.pos_wrapped, stage);
            else rc = rama_forward_stage_devtok(p->ctx, cfg, w, st, rank == 0 ? tok_dev[t.seq] : nullptr, t.pos_wrapped, stage);
            if (rc) return rc;
            if (t.samples) {      // Device::sample (cpu.rs:155-179), result stays on the device
                rc = plan->temperature == 0.0f ? rama_argmax_dev(p->ctx, st->logits, (size_t)cfg->vocab_size, tok_dev[t.seq])
                                               : rama_sample_topp_dev(p->ctx, st->logits, (size_t)cfg->vocab_size, plan->temperature, plan->topp, plan->u, tok_dev[t.seq]);
                if (rc) return rc;
                if (world == 1 && plan->out_tokens_dev &&
                    hipMemcpyAsync(plan->out_tokens_dev + (size_t)t.seq * plan->n_pos + t.pos, tok_dev[t.seq], sizeof(int32_t), hipMemcpyDeviceToDevice, stream) != hipSuccess)
                    return bad(RAMA_EIO, "rama_pipe_run_ticks: token copy failed");
            }
        }
        if (world == 1) continue;
        const float* sx = t.send_kind == RAMA_PIPE_X ? states[t.send_seq].x : nullptr;
        const int32_t* stok = t.send_kind == RAMA_PIPE_TOKEN ? tok_dev[t.send_seq] : nullptr;
        float* rx = t.recv_kind == RAMA_PIPE_X ? states[t.recv_seq].x : nullptr;
        int32_t* rtok = t.recv_kind == RAMA_PIPE_TOKEN ? tok_dev[t.recv_seq] : nullptr;
        int rc = rama_pipe_exchange(p, sx, (size_t)cfg->dim, t.send_peer, rx, (size_t)cfg->dim, t.recv_peer, stok, t.send_peer, rtok, t.recv_peer);
        if (rc) return rc;
        // rank 0 keeps the history of sampled ids (generate() prints every `next`, mod.rs:196-200)
        if (rtok && plan->out_tokens_dev &&
            hipMemcpyAsync(plan->out_tokens_dev + (size_t)t.recv_seq * plan->n_pos + t.recv_pos, rtok, sizeof(int32_t), hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return bad(RAMA_EIO, "rama_pipe_run_ticks: token copy failed");
    }
    return 0;
}

extern "C" const char* rama_pipe_last_error(void) { return g_pipe_err.c_str(); }
