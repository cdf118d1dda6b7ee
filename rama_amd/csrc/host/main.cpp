// main.cpp -- the engine CLI of engine/src/main.rs on the MI355X backend.
//   engine -m model.bin -t tokenizer.bin [-p PROMPT] [-s STEPS=255] [-r TEMPERATURE=1.0] [-l TOPP=0.9] [-o MODE]
// Same flags (main.rs:20-50), same loop (generate(), transformer/mod.rs:169-206: BOS at pos 0,
// forced prompt tokens, Device::sample afterwards, every `next` printed, no EOS stop), same
// closing line `elapsed: S.mmm s, avg tok/s: X` with X = (step - 1) / elapsed (main.rs:96-103).
// RAMA_REF_ORDER=1  (default) parity mode: every op in the reference CPU path's rounding order, logits bit-identical to cpu.rs;
//                   0 = the fast path (fused multiply-adds, tree sums), 2 = the tolerance experiment (host/engine.hpp Hip::Hip)
// RAMA_PATH=ops     forward() composed from the 1:1 Device ops (the drop-in path)
// RAMA_PATH=fused   (default) rama_forward: five fused launches per layer
// RAMA_PATH=chained the whole loop chained on the device, every token printed as it appears in the host-visible ring
// RAMA_WORLD=N RAMA_RANK=r RAMA_PIPE_ID_FILE=path [RAMA_DEVICE=d]
//                   layer pipeline over N processes, one GPU each (csrc/pipe.hip: RCCL send/recv of
//                   x[dim] and the sampled token id).  Rank r loads layers [r*L/N, (r+1)*L/N) only;
//                   rank 0 writes the RCCL unique id to the file, the others wait for it; rank 0 prints.
#include "engine.hpp"
#include "tokenizer.hpp"
#include "pipe_id.hpp"

#include <chrono>
#include <cstring>
#include <iostream>
#include <thread>

using namespace rama_host;

struct Args {
    std::string model, tokenizer, prompt = "", mode = "generate";
    uint16_t step = 255;
    float temperature = 1.0f, topp = 0.9f;
};

static void usage() {
    std::fprintf(stderr,
        "Usage: engine --model <MODEL> --tokenizer <TOKENIZER> [OPTIONS]\n"
        "  -m, --model <MODEL>              Path to the model checkpoint file\n"
        "  -t, --tokenizer <TOKENIZER>      Path to the model tokenizer file\n"
        "  -p, --prompt <PROMPT>            Initial prompt string [default: ]\n"
        "  -s, --step <STEP>                Number of steps to run [default: 255]\n"
        "  -r, --temperature <TEMPERATURE>  The temperature [0, inf] [default: 1]\n"
        "  -l, --topp <TOPP>                p value in top-p sampling [default: 0.9]\n"
        "  -o, --mode <MODE>                generate or chat [default: generate]\n"
        "environment: RAMA_REF_ORDER=1|0|2 parity (default) | fast | tolerance arithmetic; RAMA_PATH=fused|ops|chained; RAMA_TOPP_U=<draw>\n");
}

static Args parse(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; i++) {
        std::string k = argv[i];
        auto val = [&]() -> std::string {
            if (i + 1 >= argc) { usage(); std::exit(2); }
            return argv[++i];
        };
        if (k == "-m" || k == "--model") a.model = val();
        else if (k == "-t" || k == "--tokenizer") a.tokenizer = val();
        else if (k == "-p" || k == "--prompt") a.prompt = val();
        else if (k == "-s" || k == "--step") { long v = std::stol(val()); if (v < 0 || v > 65535) { usage(); std::exit(2); } a.step = (uint16_t)v; }
        else if (k == "-r" || k == "--temperature") a.temperature = std::stof(val());
        else if (k == "-l" || k == "--topp") a.topp = std::stof(val());
        else if (k == "-o" || k == "--mode") a.mode = val();
        else { usage(); std::exit(2); }
    }
    if (a.model.empty() || a.tokenizer.empty()) { usage(); std::exit(2); }
    return a;
}

// ---- RAMA_WORLD > 1: this process is one stage of the layer pipeline
static int run_pipeline_stage(const Args& args, int world, int rank) {
    const char* id_file = std::getenv("RAMA_PIPE_ID_FILE");
    if (!id_file) { std::fprintf(stderr, "RAMA_WORLD > 1 needs RAMA_PIPE_ID_FILE (a path every rank can read)\n"); return 2; }
    const char* dev_env = std::getenv("RAMA_DEVICE");
    Hip device(dev_env ? std::atoi(dev_env) : rank);
    std::ifstream rd(args.model, std::ios::binary);
    if (!rd) { std::fprintf(stderr, "couldn't open %s\n", args.model.c_str()); return 1; }
    const Config config = Config::from_file(rd);
    rd.close();
    const int L = (int)config.n_layers, base = L / world, rem = L % world;
    const int lo = rank * base + std::min(rank, rem), hi = lo + base + (rank < rem ? 1 : 0);
    const rama_stage stage{lo, hi, rank == 0, rank == world - 1};
    rama_model* model = nullptr;
    ck(rama_model_load_stage(device.ctx, args.model.c_str(), &stage, &model), "rama_model_load_stage");
    rama_config c = config.c();
    rama_weights w{};
    ck(rama_model_weights(model, &w), "rama_model_weights");
    rama_run_state st{};
    ck(rama_state_create(device.ctx, &c, hi - lo, &st), "rama_state_create");
    // the communicator: rank 0 publishes the id through RAMA_PIPE_ID_FILE, tagged with RAMA_PIPE_RUN_ID (host/pipe_id.hpp)
    unsigned char id[RAMA_PIPE_ID_BYTES];
    const char* run_id = std::getenv("RAMA_PIPE_RUN_ID");
    if (world > 1 && (!run_id || !*run_id)) {
        // without a run id a rank > 0 can pick up the id file of an EARLIER launch before rank 0 has cleared it
        std::fprintf(stderr, "RAMA_WORLD > 1 needs RAMA_PIPE_RUN_ID: the same string for every rank of this launch, a new one per launch\n");
        return 2;
    }
    if (rank == 0) {
        rama_host::pipe_id_prepare(id_file);                 // a file left by an earlier run must not meet this run's readers
        ck(rama_pipe_unique_id(id), "rama_pipe_unique_id");
        if (!rama_host::pipe_id_publish(id_file, run_id, id, sizeof id)) { std::fprintf(stderr, "cannot write %s\n", id_file); return 1; }
    } else if (!rama_host::pipe_id_wait(id_file, run_id, id, sizeof id, 60000)) {
        std::fprintf(stderr, "rank %d: no unique id of this run in %s after 60 s\n", rank, id_file);
        return 1;
    }
    rama_pipe* pipe = nullptr;
    ck(rama_pipe_create(device.ctx, id, rank, world, &pipe), "rama_pipe_create");
    if (rank == 0) rama_host::pipe_id_remove(id_file);       // every rank has joined, so every rank has read it
    if (!std::getenv("RAMA_PIPE_EAGER")) ck(rama_set_graph_mode(device.ctx, 1), "rama_set_graph_mode");   // a stage pass = one hipGraph replay

    Tokenizer tokenizer;
    std::vector<size_t> prompt_tokens;
    try {
        tokenizer = Tokenizer::from_file(args.tokenizer, config.vocab_size);
        if (!args.prompt.empty()) prompt_tokens = tokenizer.encode(args.prompt);
    } catch (const std::exception& e) { std::fprintf(stderr, "panic: %s\n", e.what()); return 101; }
    const size_t steps = args.step;
    if (steps > config.seq_len) { std::fprintf(stderr, "step %zu exceeds the checkpoint's seq_len %zu\n", steps, config.seq_len); return 1; }
    const auto start = std::chrono::steady_clock::now();
    std::vector<int32_t> pt(prompt_tokens.begin(), prompt_tokens.end());
    float* tok_word = nullptr; float* hist = nullptr;
    ck(rama_alloc_f32(device.ctx, 1, &tok_word), "rama_alloc_f32");
    ck(rama_alloc_f32(device.ctx, steps ? steps : 1, &hist), "rama_alloc_f32");
    int32_t* tok_dev[1] = {reinterpret_cast<int32_t*>(tok_word)};
    rama_pipe_plan plan{};
    plan.n_seq = 1; plan.n_pos = (int32_t)steps; plan.wrap = 0; plan.prompt = pt.data(); plan.n_prompt = (int32_t)pt.size();
    plan.temperature = args.temperature; plan.topp = args.topp; plan.u = device.topp_draw;
    plan.out_tokens_dev = rank == 0 ? reinterpret_cast<int32_t*>(hist) : nullptr;
    if (steps) ck(rama_pipe_run_ticks(pipe, &c, &w, &st, tok_dev, &stage, &plan, 0, rama_pipe_total_ticks(pipe, &plan)), "rama_pipe_run_ticks");
    ck(rama_sync(device.ctx), "rama_sync");
    if (rank == 0) {
        std::vector<float> raw(steps ? steps : 1);
        ck(rama_download_f32(device.ctx, hist, steps ? steps : 1, raw.data()), "rama_download_f32");
        const int32_t* sampled = reinterpret_cast<const int32_t*>(raw.data());
        for (size_t pos = 0; pos < steps; pos++) {          // mod.rs:190-200: the forced prompt token, else the sample
            const size_t next = pos < prompt_tokens.size() ? prompt_tokens[pos] : (size_t)sampled[pos];
            std::cout << decode(tokenizer.vocab[next]);
        }
        std::cout.flush();
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        std::printf("\n--------------------------------\n");
        std::printf("elapsed: %lld.%03lld s, avg tok/s: %g\n", (long long)elapsed, (long long)(elapsed * 1000) % 1000,
                    (double)((float)(args.step - 1) / (float)elapsed));
    }
    rama_pipe_destroy(pipe);
    rama_free(device.ctx, tok_word); rama_free(device.ctx, hist);
    rama_state_free(device.ctx, &st);
    rama_model_free(device.ctx, model);
    return 0;
}

int main(int argc, char** argv) {
    const Args args = parse(argc, argv);
    if (const char* we = std::getenv("RAMA_WORLD")) {
        const int world = std::atoi(we), rank = std::getenv("RAMA_RANK") ? std::atoi(std::getenv("RAMA_RANK")) : 0;
        if (world > 1 || std::getenv("RAMA_PIPE_ID_FILE")) {
            if (world < 1 || rank < 0 || rank >= world) { std::fprintf(stderr, "bad RAMA_WORLD / RAMA_RANK\n"); return 2; }
            return run_pipeline_stage(args, world, rank);
        }
    }
    const char* path_env = std::getenv("RAMA_PATH");
    const std::string path = path_env ? path_env : "fused";

    std::ifstream rd(args.model, std::ios::binary);
    if (!rd) { std::fprintf(stderr, "couldn't open %s\n", args.model.c_str()); return 1; }
    const Config config = Config::from_file(rd);                          // main.rs:68
    Hip device(0);                                                        // main.rs:73
    TransformerWeights weights = weights_from_file(rd, config, device);   // main.rs:76-78
    RunState state = run_state_from_config(config, device);               // main.rs:79-82
    const TransformerWeightsView wv = TransformerWeightsView::from_gpu_ws(weights);
    RunStateView rsv = RunStateView::from_rs(state);
    Tokenizer tokenizer;
    try { tokenizer = Tokenizer::from_file(args.tokenizer, config.vocab_size); }
    catch (const std::exception& e) { std::fprintf(stderr, "panic: %s\n", e.what()); return 101; }

    const size_t steps = args.step;
    if (steps > config.seq_len) {
        std::fprintf(stderr, "step %zu exceeds the checkpoint's seq_len %zu (the reference does not check and overruns its cache)\n", steps, config.seq_len);
        return 1;
    }
    const auto start = std::chrono::steady_clock::now();                  // main.rs:96 (load excluded)

    std::vector<size_t> prompt_tokens;
    try { if (!args.prompt.empty()) prompt_tokens = tokenizer.encode(args.prompt); }   // mod.rs:180
    catch (const std::exception& e) { std::fprintf(stderr, "panic: %s\n", e.what()); return 101; }

    try {
        if (path == "chained") {
            // the whole generate() loop on the device, argmax or top-p (no per-token host round trip)
            std::vector<int32_t> pt(prompt_tokens.begin(), prompt_tokens.end()), out(steps ? steps : 1);
            rama_config c = config.c(); rama_weights w = wv.c(); rama_run_state s = rsv.c();
            // every token printed as soon as the device has produced it (mod.rs:196-200 prints inside the loop), the loop
            // itself chained on the device: the host only watches the ring (rama_generate_stream)
            struct Sink { const Tokenizer* tok; } sink{&tokenizer};
            ck(rama_generate_stream(device.ctx, &c, &w, &s, pt.data(), (int)pt.size(), (int)steps, args.temperature, args.topp,
                                    device.topp_draw,
                                    [](void* user, int, int32_t token) {
                                        std::cout << decode(static_cast<Sink*>(user)->tok->vocab[(size_t)token]);
                                        std::cout.flush();
                                    }, &sink, out.data()), "rama_generate_stream");
        } else {
            size_t token = 1, pos = 0;                                    // mod.rs:182-183
            while (pos < steps) {
                if (path == "ops") forward(config, wv, rsv, token, pos, device);
                else forward_fused(config, wv, rsv, token, pos, device);
                size_t next = pos < prompt_tokens.size() ? prompt_tokens[pos]
                                                         : device.sample(config, rsv, args.temperature, args.topp);
                std::cout << decode(tokenizer.vocab[next]);               // mod.rs:196-200 (inside the timed region)
                std::cout.flush();
                token = next;
                pos += 1;
            }
        }
    } catch (const std::exception& e) { std::fprintf(stderr, "\npanic: %s\n", e.what()); return 101; }

    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    std::printf("\n--------------------------------\n");
    std::printf("elapsed: %lld.%03lld s, avg tok/s: %g\n", (long long)elapsed, (long long)(elapsed * 1000) % 1000,
                (double)((float)(args.step - 1) / (float)elapsed));
    return 0;
}
