// pipe_id_tool -- the unique-id file handshake of host/pipe_id.hpp without a GPU (tests/test_pipe_id_file.py):
//   pipe_id_tool publish <file> <run id> <fill byte>          what rank 0 does (prepare, then publish 128 bytes of <fill byte>)
//   pipe_id_tool wait    <file> <run id> <timeout ms>         what the other ranks do; prints the id's first byte
//   pipe_id_tool remove  <file>
#include "pipe_id.hpp"
#include <cstdlib>
int main(int argc, char** argv) {
    using namespace rama_host;
    if (argc < 3) return 2;
    const std::string cmd = argv[1];
    unsigned char id[128];
    if (cmd == "publish" && argc >= 5) {
        pipe_id_prepare(argv[2]);
        std::memset(id, std::atoi(argv[4]), sizeof id);
        return pipe_id_publish(argv[2], argv[3], id, sizeof id) ? 0 : 1;
    }
    if (cmd == "wait" && argc >= 5) {
        if (!pipe_id_wait(argv[2], argv[3], id, sizeof id, std::atoi(argv[4]))) return 3;
        std::printf("%d\n", (int)id[0]);
        return 0;
    }
    if (cmd == "remove") { pipe_id_remove(argv[2]); return 0; }
    return 2;
}
