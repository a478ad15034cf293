"""N > 1 path on CPU: two ranks over gloo shard the bins of ONE archive, exchange only the block sizes
(all_gather) and each writes its own blocks at the derived offsets -- the result must equal the golden
single-writer archive.  Uses the test-only host emulation of the kernels (no GPU in this container)."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT, knobs_from_flags, manifest
from test_host import assert_same_archive, emu_lib  # noqa: F401  (fixture)

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
import fastore_amd
from fastore_amd import shard
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=int(sys.argv[4]))
lib = fastore_amd.load_library(os.path.join(sys.argv[1], "build", "libfastore_emu.so"))
knobs = eval(sys.argv[7])
with fastore_amd.Packer(lib=lib, host_threads=2, rank=dist.get_rank(), world_size=dist.get_world_size(), **knobs) as p:
    shard.pack_sharded(p, sys.argv[5], sys.argv[6], dist)
    open(sys.argv[6] + ".bins%d" % dist.get_rank(), "w").write("%d" % p.stats()["bins"])
dist.destroy_process_group()
'''


@pytest.mark.parametrize("name", ["se_lossless", "pe_lossless"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pack_equals_single_writer(emu_lib, tmp_path, name, world):
    flags = [m for m in manifest() if m[0] == name][0][2]
    script = tmp_path / "w.py"; script.write_text(WORKER)
    port = str(29500 + (os.getpid() + world) % 400)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r), str(world), os.path.join(GOLDEN, name + ".in"),
                               str(tmp_path / "o"), repr(knobs_from_flags(flags))]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    assert_same_archive(str(tmp_path / "o"), os.path.join(GOLDEN, name + ".ref"))
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]
    # every rank packed a share of the bins (dealt up front: longest stream first), every bin exactly once (the archive above)
    got = [int(open(str(tmp_path / "o") + ".bins%d" % r).read()) for r in range(world)]
    assert all(b > 0 for b in got), got


WORKER_SET = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
import fastore_amd
from fastore_amd import shard
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=int(sys.argv[4]))
lib = fastore_amd.load_library(os.path.join(sys.argv[1], "build", "libfastore_emu.so"))
knobs = eval(sys.argv[7])
ins = sys.argv[5].split(","); outs = sys.argv[6].split(",")
with fastore_amd.Packer(lib=lib, host_threads=2, rank=dist.get_rank(), world_size=dist.get_world_size(), **knobs) as p:
    shard.pack_sharded_set(p, ins, outs, dist)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_set_of_libraries_equals_single_writers(emu_lib, tmp_path, world):
    # the N > 1 bench path: a SET of libraries (here the same golden library three times and a bin-stage flavoured one)
    # bin-sharded over the ranks in ONE device pipeline, one all-reduce over the concatenated size tables
    names = ["se_lossless", "se_c0", "se_lossless"]
    flags = [m for m in manifest() if m[0] == "se_lossless"][0][2]
    script = tmp_path / "w.py"; script.write_text(WORKER_SET)
    port = str(29900 + (os.getpid() + world) % 90)
    ins = ",".join(os.path.join(GOLDEN, n + ".in") for n in names); outs = ",".join(str(tmp_path / ("o%d" % i)) for i in range(len(names)))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r), str(world), ins, outs, repr(knobs_from_flags(flags))]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    for i, n in enumerate(names):
        assert_same_archive(str(tmp_path / ("o%d" % i)), os.path.join(GOLDEN, n + ".ref"))
