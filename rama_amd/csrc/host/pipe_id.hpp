// pipe_id.hpp -- how the stages of a pipeline started by hand (engine with RAMA_WORLD / RAMA_RANK) get the RCCL unique
// id from rank 0: a file every rank can read (RAMA_PIPE_ID_FILE).
//
// The file is [32 bytes run id][id bytes].  The run id is RAMA_PIPE_RUN_ID, the same string for every rank of one launch
// (e.g. "$$-$(date +%s)" in the launching shell): a reader accepts only a complete file that carries ITS run id, so the
// file of an earlier run -- or of a run that crashed before cleaning up -- is never taken for the current one.  Rank 0
// removes whatever is at the path before asking RCCL for an id, writes to a temporary name and renames (readers never see
// a partial file), and removes the file again once its communicator exists (every rank has read it by then).  Without
// RAMA_PIPE_RUN_ID the run id is empty and only the removals protect a reused path.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <unistd.h>

namespace rama_host {

constexpr size_t kPipeRunIdBytes = 32;

inline void pipe_id_remove(const char* path) { ::unlink(path); }

inline void pipe_run_id(const char* run_id, char (&out)[kPipeRunIdBytes]) {
    std::memset(out, 0, sizeof out);
    if (run_id) std::strncpy(out, run_id, sizeof out - 1);
}

// rank 0, BEFORE the id exists: nothing stale may sit at the path while the id is being made
inline void pipe_id_prepare(const char* path) { pipe_id_remove(path); pipe_id_remove((std::string(path) + ".tmp").c_str()); }

inline bool pipe_id_publish(const char* path, const char* run_id, const unsigned char* id, size_t n) {
    char rid[kPipeRunIdBytes];
    pipe_run_id(run_id, rid);
    const std::string tmp = std::string(path) + ".tmp";
    {
        std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
        if (!f) return false;
        f.write(rid, sizeof rid);
        f.write(reinterpret_cast<const char*>(id), (std::streamsize)n);
        if (!f) return false;
    }
    return std::rename(tmp.c_str(), path) == 0;
}

// ranks > 0: poll until a complete file with this launch's run id is there
inline bool pipe_id_wait(const char* path, const char* run_id, unsigned char* id, size_t n, int timeout_ms) {
    char want[kPipeRunIdBytes], got[kPipeRunIdBytes];
    pipe_run_id(run_id, want);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        {
            std::ifstream f(path, std::ios::binary);
            if (f && f.read(got, sizeof got) && f.read(reinterpret_cast<char*>(id), (std::streamsize)n) && !std::memcmp(got, want, sizeof got))
                return true;
        }
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
}

}  // namespace rama_host
