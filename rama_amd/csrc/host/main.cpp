// main.cpp -- the engine CLI of engine/src/main.rs on the MI355X backend.
//   engine -m model.bin -t tokenizer.bin [-p PROMPT] [-s STEPS=255] [-r TEMPERATURE=1.0] [-l TOPP=0.9] [-o MODE]
// Same flags (main.rs:20-50), same loop (generate(), transformer/mod.rs:169-206: BOS at pos 0,
// forced prompt tokens, Device::sample afterwards, every `next` printed, no EOS stop), same
// closing line `elapsed: S.mmm s, avg tok/s: X` with X = (step - 1) / elapsed (main.rs:96-103).
// RAMA_PATH=ops     forward() composed from the 1:1 Device ops (the drop-in path)
// RAMA_PATH=fused   (default) rama_forward: five fused launches per layer
// RAMA_PATH=chained temperature 0 only: the whole loop chained on the device, text printed at the end
#include "engine.hpp"
#include "tokenizer.hpp"

#include <chrono>
#include <cstring>
#include <iostream>

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
        "  -o, --mode <MODE>                generate or chat [default: generate]\n");
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

int main(int argc, char** argv) {
    const Args args = parse(argc, argv);
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
            ck(rama_generate(device.ctx, &c, &w, &s, pt.data(), (int)pt.size(), (int)steps, args.temperature, args.topp,
                             device.topp_draw, out.data()), "rama_generate");
            for (size_t i = 0; i < steps; i++) std::cout << decode(tokenizer.vocab[(size_t)out[i]]);
            std::cout.flush();
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
