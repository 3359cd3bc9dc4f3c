"""fq_index_build must write byte-identical index files to the ones the reference built (golden), and the C-ABI
library must export every symbol include/fastquick_amd.h declares.  No GPU needed."""
import filecmp
import os
import re
import shutil

import numpy as np
import pytest

import golden_util
from fastquick_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = api.load_library()          # the HIP product library; loads without a GPU, computes nothing here
    hdr = open(os.path.join(ROOT, "include", "fastquick_amd.h")).read()
    declared = set(re.findall(r"\b(fq_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for sym in sorted(declared):
        assert hasattr(lib, sym), "libfastquick_amd.so does not export %s" % sym
    assert set(api.EXPORTS) <= declared


def test_host_cpus_respects_the_cgroup_quota():
    """fq_host_cpus(): the hardware threads the process sees, cut to its cgroup's CPU quota when there is one (what the library sizes a
    call's host threads by)."""
    lib = api.load_library()
    n = int(lib.fq_host_cpus())
    assert 1 <= n <= (os.cpu_count() or 1)
    quota = None
    if os.path.exists("/sys/fs/cgroup/cpu.max"):
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = -(-int(q) // int(period))
    else:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                quota = -(-q // period)
        except OSError:
            pass
    if quota is not None:
        assert n == max(1, min(os.cpu_count() or 1, quota))


@pytest.mark.parametrize("tag", [t for t in golden_util.case_tags() if t != "wide"])      # (`wide`: 1,292 identical flanks back to back, the worst case of the
#   builder's plain comparison sort -- an offline tool -- at two minutes here; the other ten cases cover the builder)
def test_index_builder_matches_reference_files(tag, golden_cases, tmp_path):
    g = golden_cases[tag]
    mine = str(tmp_path / "mine.FASTQuick.fa")
    shutil.copy(g["prefix"], mine)
    api.build_index(mine, write_rollhash=False)
    for ext in golden_util.INDEX_EXT:
        assert filecmp.cmp(g["prefix"] + ext, mine + ext, shallow=False), "%s differs from the reference-built file" % ext


def test_no_device_is_a_loud_error(golden_cases):
    """Without a HIP device the product must fail, not fall back (this container has no GPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.FastquickError):
        api.Index(golden_cases["basic"]["prefix"], device=0)


def test_host_cpus_follow_the_ranks_of_a_node():
    """fq_host_cpus(): the CPUs the process may use -- its cgroup quota -- divided among the ranks of a node (torchrun's LOCAL_WORLD_SIZE),
    or stated outright (FASTQUICK_HOST_CPUS): eight ranks must not each size their host threads for the whole allowance."""
    import subprocess
    import sys
    code = "import sys; sys.path.insert(0, %r); from fastquick_amd import api; print(api.load_library().fq_host_cpus())" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_WORLD_SIZE", "FASTQUICK_HOST_CPUS")}
    whole = int(subprocess.check_output([sys.executable, "-c", code], env=env))
    assert whole >= 1
    if whole >= 2:
        half = int(subprocess.check_output([sys.executable, "-c", code], env=dict(env, LOCAL_WORLD_SIZE="2")))
        assert half == max(1, whole // 2)
    one = int(subprocess.check_output([sys.executable, "-c", code], env=dict(env, FASTQUICK_HOST_CPUS="1", LOCAL_WORLD_SIZE="8")))
    assert one == 1


def bitmap_case(tag, golden_cases, lib, tmp_path, monkeypatch):
    """The six filter tables as fq_index_load leaves them when there is neither a .rollhash nor a list of bits: from the reference's 32-mers,
    entered by the device out of the 2-bit reference in HBM (default) or listed by the host from the FASTA (FASTQUICK_HOST_BITMAPS=1: what a
    reference with letters other than ACGT takes) -- both exactly the bits of the reference's own tables (tests/golden/<case>/rollhash_bits.npz:
    BwtIndexer::AddSeq2HashCore over every record and its reverse complement, both alleles at the middle position, src/BwtIndexer.cpp:96-160)."""
    g = golden_cases[tag]
    pre = str(tmp_path / "ref.FASTQuick.fa")
    shutil.copy(g["prefix"], pre)
    for ext in golden_util.INDEX_EXT:
        shutil.copy(g["prefix"] + ext, pre + ext)
    z = np.load(os.path.join(golden_util.GOLD, tag, "rollhash_bits.npz"))
    want = [np.cumsum(z["t%d" % t].astype(np.int64)).astype(np.uint32) for t in range(6)]
    for host in (False, True):
        if host:
            monkeypatch.setenv("FASTQUICK_HOST_BITMAPS", "1")
        else:
            monkeypatch.delenv("FASTQUICK_HOST_BITMAPS", raising=False)
        ix = api.Index(pre, lib=lib)
        for t in range(6):
            got = ix.bitmap_bits(t)
            assert np.array_equal(got, want[t]), "table %d (%s): %d bits against the reference's %d" % (t, "host" if host else "device", len(got), len(want[t]))
        ix.close()


@pytest.mark.parametrize("tag", ["basic", "wide"])
def test_filter_tables_from_the_reference_are_the_references_tables(tag, golden_cases, tmp_path, monkeypatch):
    import subprocess
    emu = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "libfq_emu.so"])
    bitmap_case(tag, golden_cases, api.load_library(os.environ.get("FQ_EMU_LIB") or os.path.join(emu, "libfq_emu.so")), tmp_path, monkeypatch)
