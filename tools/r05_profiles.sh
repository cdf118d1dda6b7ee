set -e
python tools/collect_profiles.py --round 5 --out gpurun_out/profiles --mode parity > gpurun_out/r05_collect_parity.log 2>&1
python tools/collect_profiles.py --round 5 --out gpurun_out/profiles --mode fast > gpurun_out/r05_collect_fast.log 2>&1
for a in "llama2-7B 60 600 0 parity" "llama2-7B 40 300 1 parity" "stories110M 30 1024 0 parity" "stories15M 30 256 1"; do python tools/soak.py $a >> gpurun_out/r05_soak.txt 2>&1; done
cat gpurun_out/r05_soak.txt | cut -c1-300; ls gpurun_out/profiles
