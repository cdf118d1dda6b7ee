"""CPU: the unique-id file of a hand-started pipeline (csrc/host/pipe_id.hpp, used by the engine CLI's stage mode).
A reader must take only a complete file that carries ITS launch's run id; rank 0 clears the path before it publishes;
the same path serves one launch after another, including after a crash that left a file behind."""
import subprocess
import time
from pathlib import Path

import pytest

TOOL = Path(__file__).resolve().parent.parent / "rama_amd" / "bin" / "pipe_id_tool"


def run(*a, timeout=30):
    return subprocess.run([str(TOOL), *map(str, a)], capture_output=True, text=True, timeout=timeout)


@pytest.fixture(autouse=True)
def _built():
    if not TOOL.exists():
        import __graft_entry__
        __graft_entry__.build()
    assert TOOL.exists()


def test_two_launches_against_one_path(tmp_path):
    f = tmp_path / "pipe.id"
    for launch, fill in (("run-A", 11), ("run-B", 22)):
        reader = subprocess.Popen([str(TOOL), "wait", str(f), launch, "10000"], stdout=subprocess.PIPE, text=True)
        time.sleep(0.2)                                   # the reader is polling before rank 0 has published
        assert run("publish", f, launch, fill).returncode == 0
        out, _ = reader.communicate(timeout=20)
        assert reader.returncode == 0 and int(out) == fill
        # the first launch "crashes" here: its file stays behind for the second one to find


def test_stale_file_of_another_run_is_never_accepted(tmp_path):
    f = tmp_path / "pipe.id"
    assert run("publish", f, "old-run", 33).returncode == 0
    r = run("wait", f, "new-run", 400)
    assert r.returncode == 3 and r.stdout == ""           # timed out rather than joining a dead root
    reader = subprocess.Popen([str(TOOL), "wait", str(f), "new-run", "10000"], stdout=subprocess.PIPE, text=True)
    time.sleep(0.2)
    assert run("publish", f, "new-run", 44).returncode == 0
    out, _ = reader.communicate(timeout=20)
    assert reader.returncode == 0 and int(out) == 44


def test_partial_file_is_not_an_id(tmp_path):
    f = tmp_path / "pipe.id"
    f.write_bytes(b"run-C".ljust(32, b"\0") + b"\x05" * 100)       # 28 bytes short
    assert run("wait", f, "run-C", 300).returncode == 3


def test_remove(tmp_path):
    f = tmp_path / "pipe.id"
    assert run("publish", f, "", 1).returncode == 0 and f.exists()
    assert run("remove", f).returncode == 0 and not f.exists()
