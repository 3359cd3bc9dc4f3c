"""`FASTQuick_amd align --devices LIST`: the lines of a --fq_list dealt over several devices from ONE process (no Python, no torch) --
one copy of the index, one alignment context chain and one shard consumer per device; SAM text / BAM records appended and the
StatCollector segments merged (fq_qc_merge) in input order.  The output must be the one-device run's, byte for byte, and that is the
REAL reference's two-pair run (tests/golden/qc/ref_fqlist.*; src/BwtMapper.cpp:232-262, src/StatCollector.h:46-62).
CPU tier: the front end over the host-loop library, whose "devices" are virtual; GPU tier: two workers on one device, and -- where the
box has them -- two devices."""
import os
import subprocess

import pytest

import golden_util
from test_qc_consumer import QC_FILES, qc_bytes

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)




def bam_payload(path):
    """what a BAM file inflates to, every BGZF member checked on the way (how the record stream was cut into members, and by which compressor --
    zlib on the host, or the device's (csrc/fq_deflate.h) -- is not part of the file's content)"""
    import struct
    import zlib
    blob, out, at = open(path, "rb").read(), [], 0
    while at < len(blob):
        assert blob[at:at + 4] == b"\x1f\x8b\x08\x04" and blob[at + 12:at + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", blob, at + 16)[0] + 1
        data = zlib.decompress(blob[at + 18:at + bsize - 8], -15)
        crc, isz = struct.unpack("<II", blob[at + bsize - 8:at + bsize])
        assert zlib.crc32(data) == crc and len(data) == isz
        out.append(data)
        at += bsize
    assert out and out[-1] == b"", "the end-of-file member"
    return b"".join(out)

def run_list(exe, g, tmp, tag, devices=None, sam=True):
    halves = golden_util.split_halves(g, str(tmp))
    lst = os.path.join(str(tmp), "two_pairs.list")
    with open(lst, "w") as fh:
        fh.write("# the two halves of the case's FASTQ pair\n" + "".join("%s\t%s\n" % h for h in halves))
    prefix = g["prefix"][:-len(".FASTQuick.fa")]
    with open(g["prefix"] + ".param", "w") as fh:
        fh.write("REFERENCE_PATH\t%s\nTARGET_REGION_PATH\tEmpty\nDBSNP_VCF_PATH\tEmpty\nNUM_VAR_LONG\t4\nNUM_VAR_SHORT\t36\n"
                 "SHORT_FLANK_LENGTH\t250\nLONG_FLANK_LENGTH\t1000\n" % os.path.join(g["dir"], "genome"))
    out = os.path.join(str(tmp), tag)
    cmd = [exe, "align", "--index_prefix", prefix, "--fq_list", lst, "--out_prefix", out, "--batch_pairs", str(g["batch"]),
           "--chunk_pairs", str(2 * g["batch"]), "--q", str(g["trim_qual"]), "--read_len", str(g["qc_read_len"])]
    if sam:
        cmd.append("--sam_out")
    if devices:
        cmd += ["--devices", devices]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode == 0, run.stderr.decode(errors="replace")[-3000:]
    return run, out


def check_against_one_device(exe, g, tmp, devices):
    one, out1 = run_list(exe, g, tmp, "one")
    many, outn = run_list(exe, g, tmp, "many", devices=devices)
    assert many.stderr.count(b"takes line") == 2, "both lines of the list must have been dealt to a worker"
    assert many.stdout == one.stdout
    assert many.stdout == open(os.path.join(g["dir"], "ref_fqlist.sam"), "rb").read(), "SAM text of the reference's two-pair run"
    for f in QC_FILES:
        assert qc_bytes(outn + "." + f).replace(outn.encode(), b"OUT") == qc_bytes(out1 + "." + f).replace(out1.encode(), b"OUT"), f
        if f not in ("Summary", "FASTQ.csv"):    # (genome size: the three contigs of the .fai here, one genome in the golden run; file names)
            assert qc_bytes(outn + "." + f) == qc_bytes(os.path.join(g["dir"], "ref_fqlist.qc." + f)), f
    left = [n for n in os.listdir(str(tmp)) if ".part" in n or ".worker" in n]
    assert not left, "part files and worker files must be gone: %s" % left
    one, out1 = run_list(exe, g, tmp, "one_bam", sam=False)
    many, outn = run_list(exe, g, tmp, "many_bam", devices=devices, sam=False)
    assert bam_payload(outn + ".bam") == bam_payload(out1 + ".bam"), "the BAM file does not depend on how its records were produced"


def run_pair(exe, g, tmp, tag, devices=None, sam=True):
    prefix = g["prefix"][:-len(".FASTQuick.fa")]
    with open(g["prefix"] + ".param", "w") as fh:
        fh.write("REFERENCE_PATH\t%s\nTARGET_REGION_PATH\tEmpty\nDBSNP_VCF_PATH\tEmpty\nNUM_VAR_LONG\t4\nNUM_VAR_SHORT\t36\n"
                 "SHORT_FLANK_LENGTH\t250\nLONG_FLANK_LENGTH\t1000\n" % os.path.join(g["dir"], "genome"))
    out = os.path.join(str(tmp), tag)
    cmd = [exe, "align", "--index_prefix", prefix, "--fastq_1", g["fq1"], "--fastq_2", g["fq2"], "--out_prefix", out, "--batch_pairs", str(g["batch"]),
           "--chunk_pairs", str(g["batch"]), "--q", str(g["trim_qual"]), "--read_len", str(g["qc_read_len"])]
    if sam:
        cmd.append("--sam_out")
    if devices:
        cmd += ["--devices", devices]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode == 0, run.stderr.decode(errors="replace")[-3000:]
    return run, out


def check_one_pair_sharded(exe, g, tmp, devices):
    """ONE FASTQ pair over several devices (chunks dealt round-robin, the drand48 stream / last_ii / (k,l) cache handed from context to
    context around each call's serial part): the output of the one-device run, which is the reference's for the case."""
    assert g["n_pairs"] > 2 * g["batch"], "the case must span several chunks"
    one, out1 = run_pair(exe, g, tmp, "one")
    many, outn = run_pair(exe, g, tmp, "many", devices=devices)
    assert b"mapping on 2 devices" in many.stderr
    assert many.stdout == one.stdout == open(g["sam"], "rb").read(), "SAM text of the reference's run of the pair"
    for f in QC_FILES:
        assert qc_bytes(outn + "." + f).replace(outn.encode(), b"OUT") == qc_bytes(out1 + "." + f).replace(out1.encode(), b"OUT"), f
        if f not in ("Summary", "FASTQ.csv"):
            assert qc_bytes(outn + "." + f) == qc_bytes(os.path.join(g["dir"], "ref.qc." + f)), f
    one, out1 = run_pair(exe, g, tmp, "one_bam", sam=False)
    many, outn = run_pair(exe, g, tmp, "many_bam", devices=devices, sam=False)
    assert bam_payload(outn + ".bam") == bam_payload(out1 + ".bam")


@pytest.mark.parametrize("tag", ["basic", "qc"])      # three chunks each: both workers get work, the state crosses contexts twice
def test_one_pair_sharded_over_two_virtual_devices(tag, golden_cases, tmp_path):
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "libfq_emu.so", "FASTQuick_emu"])
    check_one_pair_sharded(os.path.join(emu, "FASTQuick_emu"), golden_cases[tag], tmp_path, "0,1")


@pytest.mark.gpu
def test_one_pair_sharded_over_two_workers_on_one_gpu(golden_cases, tmp_path):
    check_one_pair_sharded(os.path.join(ROOT, "fastquick_amd", "bin", "FASTQuick_amd"), golden_cases["qc"], tmp_path, "0,0")


def test_fq_list_over_two_virtual_devices(golden_cases, tmp_path):
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "libfq_emu.so", "FASTQuick_emu"])
    check_against_one_device(os.path.join(emu, "FASTQuick_emu"), golden_cases["qc"], tmp_path, "0,1")


def run_list8(exe, g, tmp, tag, devices=None, sam=True):
    """the case's pair cut into eight pairs of BGZF files of unequal sizes (the lines of a --fq_list): on a GPU the device front end reads them,
    one reader set per worker"""
    from fastquick_amd import synth
    texts = [open(g[k], "rb").read().split(b"\n") for k in ("fq1", "fq2")]
    n = (len(texts[0]) - 1) // 4
    cuts = [0] + [n * k // 11 for k in (1, 2, 4, 5, 7, 8, 10)] + [n]
    lst = os.path.join(str(tmp), "eight_pairs.list")
    with open(lst, "w") as fh:
        for i in range(8):
            paths = []
            for e in range(2):
                path = os.path.join(str(tmp), "part8_%d_%d.fq.gz" % (i, e + 1))
                if not os.path.exists(path):
                    with open(path, "wb") as fo:
                        fo.write(synth.bgzf_compress(b"\n".join(texts[e][4 * cuts[i]:4 * cuts[i + 1]]) + b"\n", threads=2, level=6, member=3000))
                paths.append(path)
            fh.write("%s\t%s\n" % tuple(paths))
    prefix = g["prefix"][:-len(".FASTQuick.fa")]
    with open(g["prefix"] + ".param", "w") as fh:
        fh.write("REFERENCE_PATH\t%s\nTARGET_REGION_PATH\tEmpty\nDBSNP_VCF_PATH\tEmpty\nNUM_VAR_LONG\t4\nNUM_VAR_SHORT\t36\n"
                 "SHORT_FLANK_LENGTH\t250\nLONG_FLANK_LENGTH\t1000\n" % os.path.join(g["dir"], "genome"))
    out = os.path.join(str(tmp), tag)
    cmd = [exe, "align", "--index_prefix", prefix, "--fq_list", lst, "--out_prefix", out, "--batch_pairs", str(g["batch"]),
           "--chunk_pairs", str(2 * g["batch"]), "--q", str(g["trim_qual"]), "--read_len", str(g["qc_read_len"])]
    if sam:
        cmd.append("--sam_out")
    if devices:
        cmd += ["--devices", devices]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode == 0, run.stderr.decode(errors="replace")[-3000:]
    return run, out


def check_eight_workers(exe, g, tmp, devices, bam=True):
    one, out1 = run_list8(exe, g, tmp, "one8")
    many, outn = run_list8(exe, g, tmp, "many8", devices=devices)
    assert many.stderr.count(b"takes line") == 8, "every line of the list must have been dealt to a worker"
    assert len({ln.split(b" takes")[0] for ln in many.stderr.split(b"\n") if b"takes line" in ln}) >= 1
    assert many.stdout == one.stdout and len(one.stdout) > 1000
    for f in QC_FILES:
        assert qc_bytes(outn + "." + f).replace(outn.encode(), b"OUT") == qc_bytes(out1 + "." + f).replace(out1.encode(), b"OUT"), f
    left = [x for x in os.listdir(str(tmp)) if ".part" in x and ".fq.gz" not in x or ".worker" in x]
    assert not left, "part files and worker files must be gone: %s" % left
    if bam:
        one, out1 = run_list8(exe, g, tmp, "one8_bam", sam=False)
        many, outn = run_list8(exe, g, tmp, "many8_bam", devices=devices, sam=False)
        assert bam_payload(outn + ".bam") == bam_payload(out1 + ".bam")


def test_eight_line_fq_list_over_eight_workers_on_virtual_devices(golden_cases, tmp_path):
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "libfq_emu.so", "FASTQuick_emu"])
    check_eight_workers(os.path.join(emu, "FASTQuick_emu"), golden_cases["qc"], tmp_path, "0,1,2,3,0,1,2,3", bam=False)   # (BAM as well: the GPU tier)


@pytest.mark.gpu
def test_eight_line_fq_list_over_eight_workers_on_one_gpu(golden_cases, tmp_path):
    """--devices 0,0,0,0,0,0,0,0 on an 8-line --fq_list == the one-device output (each worker its own index copy, contexts,
    front end with its readers, consumers)"""
    check_eight_workers(os.path.join(ROOT, "fastquick_amd", "bin", "FASTQuick_amd"), golden_cases["qc"], tmp_path, "0,0,0,0,0,0,0,0")


def _hip_devices():
    import torch
    return torch.cuda.device_count()


@pytest.mark.gpu
def test_fq_list_over_two_workers_on_one_gpu(golden_cases, tmp_path):
    check_against_one_device(os.path.join(ROOT, "fastquick_amd", "bin", "FASTQuick_amd"), golden_cases["qc"], tmp_path, "0,0")


@pytest.mark.gpu
def test_fq_list_over_two_gpus(golden_cases, tmp_path):
    if _hip_devices() < 2:
        pytest.skip("one HIP device on this box")
    check_against_one_device(os.path.join(ROOT, "fastquick_amd", "bin", "FASTQuick_amd"), golden_cases["qc"], tmp_path, "0-1")


@pytest.mark.parametrize("devices,msg", [("a,b", b"bad device ordinal"), ("0-x", b"bad device ordinal"), ("0,,1", b"empty entry"), ("0,1,", b"empty entry"), ("1.5", b"bad device ordinal"),
                                         ("0,63", b"does not exist")])
def test_devices_option_is_parsed_strictly_and_checked_before_any_worker_starts(devices, msg, golden_cases, emu_cli, tmp_path):
    """`--devices a,b` / `0-x` went through atoi and became device 0; an ordinal beyond the visible devices failed inside a
    worker thread and left part files behind."""
    g = golden_cases["qc"]
    halves = golden_util.split_halves(g, str(tmp_path))
    lst = os.path.join(str(tmp_path), "two_pairs.list")
    with open(lst, "w") as fh:
        fh.write("".join("%s\t%s\n" % h for h in halves))
    out = os.path.join(str(tmp_path), "out")
    cmd = [emu_cli, "align", "--index_prefix", g["prefix"][:-len(".FASTQuick.fa")], "--fq_list", lst, "--out_prefix", out, "--sam_out", "--devices", devices]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode != 0 and msg in run.stderr, run.stderr.decode(errors="replace")[-1000:]
    left = [n for n in os.listdir(str(tmp_path)) if ".part" in n or ".worker" in n]
    assert not left, "nothing may be left behind: %s" % left
