"""CPU: bench.py --gpus N without a launcher starts its N ranks itself (spawn_ranks): one process per rank with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and a free port, rank 0's line passed through, any failure or a
timeout ends every child and gives a non-zero exit code.  Driven here with stub stages (no GPU)."""
import json
import os
import subprocess
import sys
import textwrap
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import bench  # noqa: E402


def _stub(tmp_path, body):
    p = tmp_path / "stub_rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_spawn_ranks_passes_rank0_line_and_env(tmp_path, capfd):
    stub = _stub(tmp_path, """
        import json, os, socket, sys, time
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1"
        port = int(os.environ["MASTER_PORT"])
        # a gloo-less rendezvous: rank 0 listens on the port the launcher picked, the others connect
        if r == 0:
            s = socket.socket(); s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1); s.bind(("127.0.0.1", port)); s.listen(w)
            peers = sorted(int(s.accept()[0].recv(16).decode()) for _ in range(w - 1))
            print(json.dumps({"value": 1.0, "n_gpus": w, "peers": peers, "argv": sys.argv[1:]}), flush=True)
        else:
            for _ in range(100):
                try:
                    c = socket.create_connection(("127.0.0.1", port), timeout=1); break
                except OSError:
                    time.sleep(0.05)
            c.send(str(r).encode()); c.close()
    """)
    rc = bench.spawn_ranks(["--gpus", "3", "--steps", "2"], 3, timeout=60, script=stub)
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1
    line = json.loads(out[0])
    assert line["n_gpus"] == 3 and line["peers"] == [1, 2] and line["argv"] == ["--gpus", "3", "--steps", "2"]


def test_spawn_ranks_one_failure_ends_the_others(tmp_path, capfd):
    stub = _stub(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(120)          # the survivors would hang in a collective
    """)
    t0 = time.time()
    rc = bench.spawn_ranks([], 3, timeout=60, script=stub)
    assert rc == 7 and time.time() - t0 < 30
    assert capfd.readouterr().out == ""          # no line on stdout from a failed job


def test_spawn_ranks_timeout(tmp_path):
    stub = _stub(tmp_path, "import time; time.sleep(120)")
    t0 = time.time()
    rc = bench.spawn_ranks([], 2, timeout=1.0, script=stub)
    assert rc == 124 and time.time() - t0 < 30


def test_bench_gpus_n_without_launcher_spawns_before_touching_a_gpu(tmp_path):
    """`python bench.py --gpus 2` with no RANK in the environment: [r6] on a node that shows fewer than 2 devices it says "needs 2 visible devices" and
    exits 3 BEFORE anything touches a GPU; with the device check waived it goes through spawn_ranks (here: the children fail loudly because this
    container has no GPU, and the parent reports it with a non-zero code instead of hanging)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    import torch
    cmd = [sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--rank-timeout", "120"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 3 and "needs 2 visible devices" in r.stderr and r.stdout.strip() == ""
    if not torch.cuda.is_available():
        r = subprocess.run(cmd, env=dict(env, RAMA_BENCH_SKIP_DEVICE_CHECK="1"), capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "rank" in r.stderr
        assert r.stdout.strip() == ""


def test_provenance_stamps(tmp_path):
    """[r6] bench.py stamps what it prints, and the collectors what they write, with the hash of the SOURCES librama_hip.so is built from (hipcc's output
    is not bit-reproducible, so the binary's own hash would not survive a rebuild): the hash is stable, __graft_entry__.build() records it, a stale build
    is flagged, and a profile's stamp is read from its JSON or from the .meta.json beside a CSV"""
    h = bench.source_hash()
    assert isinstance(h, str) and len(h) == 16 and h == bench.source_hash()
    info = REPO / "rama_amd" / "BUILD_INFO.json"
    if info.exists():          # written by build(); the driver builds before it tests
        st = bench.library_stamp()
        recorded = json.loads(info.read_text()).get("src_sha16")
        assert st["src_sha16"] == h and st["built_from_src_sha16"] == recorded
        assert ("stale_build" in st) == (recorded != h)
    j = tmp_path / "r99_x.json"
    j.write_text(json.dumps({"library": {"src_sha16": "0123456789abcdef"}, "rows": []}))
    assert bench.profile_stamp(j) == "0123456789abcdef"
    c = tmp_path / "r99_kernel_stats.csv"
    c.write_text("Name,Calls\n")
    assert bench.profile_stamp(c) is None                      # no sidecar: rounds 1-5
    c.with_suffix(".meta.json").write_text(json.dumps({"library": {"src_sha16": "fedcba9876543210"}}))
    assert bench.profile_stamp(c) == "fedcba9876543210"
    assert bench.profile_stamp(tmp_path / "missing.json") is None
