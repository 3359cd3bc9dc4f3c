"""`FASTQuick align` without --sam_out, as the pipeline script calls it: records to <out>.bam in genome coordinates and the QC
files of StatCollector next to it, from the C++ front end (fq_cli.cpp) through the C ABI.  Checked against the REAL reference's
SamRecords and QC files of the golden case `qc` (CPU tier: the front end linked against the host-loop library)."""
import os
import subprocess

import golden_util  # noqa: F401  (fixtures)
from test_bam_writer import check_bgzf, decode_bam
from test_qc_consumer import QC_FILES, qc_bytes

HERE = os.path.dirname(os.path.abspath(__file__))


def cli_bam_and_qc(exe, g, out, se=False):
    """se: only --fastq_1 is given (BwtMapper::SingleEndMapper), checked against the ref_se.* goldens."""
    prefix = g["prefix"][:-len(".FASTQuick.fa")]
    genome = os.path.join(g["dir"], "genome")               # <REFERENCE_PATH>.fai is the golden's genome.fai
    with open(g["prefix"] + ".param", "w") as fh:           # the 7 lines `FASTQuick index` writes (src/FASTQuick.cpp:145-151)
        fh.write("REFERENCE_PATH\t%s\nTARGET_REGION_PATH\tEmpty\nDBSNP_VCF_PATH\tEmpty\nNUM_VAR_LONG\t4\nNUM_VAR_SHORT\t36\n"
                 "SHORT_FLANK_LENGTH\t250\nLONG_FLANK_LENGTH\t1000\n" % genome)
    cmd = [exe, "align", "--index_prefix", prefix, "--fastq_1", g["fq1"]] + ([] if se else ["--fastq_2", g["fq2"]]) + ["--out_prefix", out,
           "--batch_pairs", str(g["batch"]), "--chunk_pairs", str(2 * g["batch"])]
    stem = "ref_se" if se else "ref"
    if g["trim_qual"]:
        cmd += ["--q", str(g["trim_qual"])]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode == 0, run.stderr.decode(errors="replace")[-2000:]
    assert run.stdout == b"", "no SAM text on stdout in BAM mode"
    check_bgzf(out + ".bam")
    text, _refs, recs = decode_bam(out + ".bam", sort_tags=False)
    assert text == open(os.path.join(g["dir"], stem + ".bamhdr")).read()
    want = [l.rstrip("\n").split("\t") for l in open(os.path.join(g["dir"], stem + ".bamtxt"))]
    assert recs == want
    for f in QC_FILES:
        if f in ("Summary", "FASTQ.csv"):    # (genome size: the three contigs of the .fai here, one genome in the golden run; file names)
            continue
        assert qc_bytes(out + "." + f) == qc_bytes(os.path.join(g["dir"], stem + ".qc." + f)), f
    assert os.path.getsize(out + ".Summary") > 100


def test_cli_writes_bam_and_qc_files(golden_cases, tmp_path):
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "libfq_emu.so", "FASTQuick_emu"])
    cli_bam_and_qc(os.path.join(emu, "FASTQuick_emu"), golden_cases["qc"], str(tmp_path / "cli_qc"))


def test_cli_single_end_writes_bam_and_qc_files(golden_cases, tmp_path):
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "libfq_emu.so", "FASTQuick_emu"])
    cli_bam_and_qc(os.path.join(emu, "FASTQuick_emu"), golden_cases["edge"], str(tmp_path / "cli_se"), se=True)


def test_cli_single_end_sam_text(golden_cases, tmp_path):
    """--sam_out with --fastq_1 alone: the single-end mapper's SAM text."""
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "libfq_emu.so", "FASTQuick_emu"])
    g = golden_cases["trim76"]
    prefix = g["prefix"][:-len(".FASTQuick.fa")]
    cmd = [os.path.join(emu, "FASTQuick_emu"), "align", "--index_prefix", prefix, "--fastq_1", g["fq1"], "--out_prefix", str(tmp_path / "se"), "--sam_out",
           "--batch_pairs", str(g["batch"]), "--q", str(g["trim_qual"])]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode == 0, run.stderr.decode(errors="replace")[-2000:]
    assert run.stdout == open(g["se_sam"], "rb").read()


def cli_frac_samp(exe, g, out, frac=0.8):
    """--frac_samp: the reader drops the records the reference's generator drops; SAM text and QC files of the down-sampled run."""
    prefix = g["prefix"][:-len(".FASTQuick.fa")]
    cmd = [exe, "align", "--index_prefix", prefix, "--fastq_1", g["fq1"], "--fastq_2", g["fq2"], "--out_prefix", out, "--sam_out", "--frac_samp", str(frac),
           "--batch_pairs", str(g["batch"]), "--chunk_pairs", str(2 * g["batch"]), "--q", str(g["trim_qual"])]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode == 0, run.stderr.decode(errors="replace")[-2000:]
    want = open(os.path.join(g["dir"], "ref_frac.sam"), "rb").read()
    assert 0.6 * len(open(g["sam"], "rb").read()) < len(want) < 0.95 * len(open(g["sam"], "rb").read())
    assert run.stdout == want
    for f in QC_FILES:
        if f in ("Summary", "FASTQ.csv"):
            continue
        assert qc_bytes(out + "." + f) == qc_bytes(os.path.join(g["dir"], "ref_frac.qc." + f)), f


def test_cli_frac_samp(golden_cases, tmp_path):
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "libfq_emu.so", "FASTQuick_emu"])
    cli_frac_samp(os.path.join(emu, "FASTQuick_emu"), golden_cases["qc"], str(tmp_path / "frac"))


def test_cli_76bp_pairs_with_the_default_read_len(golden_cases, emu_cli, tmp_path):
    """the command line refuses --read_len < 96, and the reference has no such flag (gap_opt_t::read_len is 151 whatever the
    reads' length): 2x76 reads -- BASELINE cfg 5's shape -- run with the default, rows sized from the first record; the reference's SAM text."""
    g = golden_cases["trim76"]
    cmd = [emu_cli, "align", "--index_prefix", g["prefix"][:-len(".FASTQuick.fa")], "--fastq_1", g["fq1"], "--fastq_2", g["fq2"], "--out_prefix", str(tmp_path / "pe76"),
           "--sam_out", "--batch_pairs", str(g["batch"]), "--q", str(g["trim_qual"])]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode == 0, run.stderr.decode(errors="replace")[-2000:]
    assert run.stdout == open(g["sam"], "rb").read()


def bgzf_copy(g, tmp, member=2500):
    """the case with its FASTQ files as BGZF: the command line then inflates and tokenises them on the device (fq_frontend_*), runs its calls on
    two contexts in turn (the stream's state handed from one to the other) and its consumers beside the next call"""
    from fastquick_amd import synth
    out = dict(g)
    for k in ("fq1", "fq2"):
        path = os.path.join(str(tmp), os.path.basename(g[k]) + ".gz")
        with open(path, "wb") as fh:
            fh.write(synth.bgzf_compress(open(g[k], "rb").read(), threads=2, level=6, member=member))
        out[k] = path
    return out


def test_cli_bgzf_input_writes_the_same_bam_and_qc_files(golden_cases, emu_cli, tmp_path):
    g = bgzf_copy(golden_cases["qc"], tmp_path)
    g["batch"] = golden_cases["qc"]["batch"]
    cli_bam_and_qc(emu_cli, g, str(tmp_path / "cli_qc_bgzf"))


def test_cli_bgzf_input_over_several_chunks_prints_the_references_sam(golden_cases, emu_cli, tmp_path):
    """`repeat`: placements are drawn from the drand48 stream, so a state that did not travel with the calls from context to context would show"""
    for tag in ("repeat", "wide", "isize"):
        g = bgzf_copy(golden_cases[tag], tmp_path)
        cmd = [emu_cli, "align", "--index_prefix", g["prefix"][:-len(".FASTQuick.fa")], "--fastq_1", g["fq1"], "--fastq_2", g["fq2"], "--out_prefix", str(tmp_path / ("sam_" + tag)),
               "--sam_out", "--batch_pairs", str(g["batch"] // 4), "--chunk_pairs", str(g["batch"] // 4)] + (["--q", str(g["trim_qual"])] if g["trim_qual"] else [])
        ref = [emu_cli, "align", "--index_prefix", g["prefix"][:-len(".FASTQuick.fa")], "--fastq_1", golden_cases[tag]["fq1"], "--fastq_2", golden_cases[tag]["fq2"], "--out_prefix", str(tmp_path / ("ref_" + tag)),
               "--sam_out", "--batch_pairs", str(g["batch"] // 4), "--chunk_pairs", str(g["batch"])] + (["--q", str(g["trim_qual"])] if g["trim_qual"] else [])
        run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert run.returncode == 0, run.stderr.decode(errors="replace")[-2000:]
        assert b"front end on the device" in run.stderr and b"read on by the host" not in run.stderr
        host = subprocess.run(ref, stdout=subprocess.PIPE, stderr=subprocess.PIPE)      # the host reader on the plain files, other chunking: the same stream
        assert host.returncode == 0 and b"front end on the device" not in host.stderr
        assert run.stdout == host.stdout, tag


import pytest  # noqa: E402


@pytest.mark.gpu
def test_cli_device_front_end_equals_the_host_reader_on_four_million_pairs(tmp_path):
    """The command line at the size it is measured at: 4,194,304 + 262,144 + 1,000 pairs of a WGS-like mix in two BGZF files -- several chunks of
    sixteen reference batches through the front end on the device, two alignment contexts in turn (the stream's state exported / imported
    between them), the consumers of one call on their own threads beside the next call -- must print the SAM text and write the QC files of
    the run that reads with the host's reader and packer (`--host_reader`: the path every parity test of round 4 went through)."""
    import hashlib
    import numpy as np
    from fastquick_amd import api, synth
    ref = synth.make_reference(n_markers=2000, n_long=200, seed=811)
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    api.build_index(pre)
    synth.write_qc_inputs(pre, ref)
    synth.write_param(pre, ref, 1000)
    with open(pre + ".genome.fa.fai", "w") as fh:
        fh.write("1\t%d\t3\t60\t61\n" % len(ref.genome))
    n1 = 1 << 20
    rb = synth.make_reads(ref, n1, on_target=0.01, seed=812, sub_rate=0.006, del_frac=0.02, ins_frac=0.02, n_rate=0.001, chimera_frac=0.01)
    fq = []
    for e in range(2):
        one = str(tmp_path / ("one_%d.fq.gz" % (e + 1)))
        synth.write_fastq_uniform(rb.seq[e], rb.qual[e], 150, one, threads=8)
        blob = open(one, "rb").read()
        eof = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00"
        body = blob[:-28] if blob.endswith(eof) else blob
        tail = str(tmp_path / ("tail_%d.fq.gz" % (e + 1)))
        synth.write_fastq_uniform(rb.seq[e][:263144], rb.qual[e][:263144], 150, tail, name_prefix=b"t", threads=8)
        path = str(tmp_path / ("big_%d.fq.gz" % (e + 1)))
        with open(path, "wb") as fo:
            for _ in range(4):
                fo.write(body)
            fo.write(open(tail, "rb").read())
        os.remove(one); os.remove(tail)
        fq.append(path)
    exe = os.path.join(os.path.dirname(HERE), "fastquick_amd", "bin", "FASTQuick_amd")
    outs = {}
    for mode, extra in (("device", []), ("host", ["--host_reader"])):
        out = str(tmp_path / mode)
        cmd = [exe, "align", "--index_prefix", pre[:-len(".FASTQuick.fa")], "--fastq_1", fq[0], "--fastq_2", fq[1], "--out_prefix", out, "--sam_out", "--read_len", "151"] + extra
        run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert run.returncode == 0, run.stderr.decode(errors="replace")[-2000:]
        if mode == "device":
            assert b"front end on the device: 4457448 pairs" in run.stderr, run.stderr.decode(errors="replace")[-1500:]
        outs[mode] = (hashlib.sha256(run.stdout).hexdigest(), len(run.stdout), {f: qc_bytes(out + "." + f).replace(out.encode(), b"OUT") for f in QC_FILES})
    assert outs["device"][1] == outs["host"][1] > 10e6 and outs["device"][0] == outs["host"][0], "SAM text"
    for f in QC_FILES:
        assert outs["device"][2][f] == outs["host"][2][f], f


@pytest.mark.parametrize("n1,n2", [(0, 0), (1, 1), (600, 300), (300, 600), (512, 256)], ids=["empty", "one_pair", "second_file_short", "first_file_short", "short_at_a_batch_boundary"])
def test_cli_degenerate_inputs_device_front_end_equals_the_host_reader(n1, n2, golden_cases, emu_cli, tmp_path):
    """no record at all, a single pair, files of unequal record counts (the command line pairs record i with record i and ends with the shorter
    file; here only that both front ends do the same): BGZF files through the device front end give the SAM text and QC files of the plain
    files through the host's reader"""
    from fastquick_amd import synth
    g = golden_cases["qc"]
    lines = [open(g[k], "rb").read().split(b"\n") for k in ("fq1", "fq2")]
    res = {}
    for mode in ("plain", "bgzf"):
        work = tmp_path / mode
        work.mkdir()
        paths = []
        for e, n in enumerate((n1, n2)):
            text = b"".join(x + b"\n" for x in lines[e][:4 * n])
            p = str(work / ("r%d.fq" % (e + 1) + (".gz" if mode == "bgzf" else "")))
            with open(p, "wb") as fh:
                fh.write(text if mode == "plain" else synth.bgzf_compress(text, threads=1, level=6, member=4000))
            paths.append(p)
        cmd = [emu_cli, "align", "--index_prefix", g["prefix"][:-len(".FASTQuick.fa")], "--fastq_1", paths[0], "--fastq_2", paths[1], "--out_prefix", str(work / "o"),
               "--sam_out", "--batch_pairs", "256", "--chunk_pairs", "256"]
        run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert run.returncode == 0, run.stderr.decode(errors="replace")[-2000:]
        assert (b"front end on the device" in run.stderr) == (mode == "bgzf")
        qc = {f: open(str(work / f), "rb").read() for f in sorted(os.listdir(str(work))) if f.startswith("o.")}
        qc["o.FASTQ.csv"] = qc["o.FASTQ.csv"].replace(b".fq.gz", b".fq")         # the one place the input files' names are written
        res[mode] = (run.stdout, qc)
    assert res["plain"][0] == res["bgzf"][0]
    assert sorted(res["plain"][1]) == sorted(res["bgzf"][1]) and len(res["plain"][1]) == 13
    for f in res["plain"][1]:
        assert res["plain"][1][f] == res["bgzf"][1][f], f


class _Load:
    """busy-loop processes beside the command line: a refusal must not depend on which thread the scheduler runs first"""
    def __init__(self, n):
        self.n = n
    def __enter__(self):
        import sys
        self.p = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(self.n)]
        return self
    def __exit__(self, *a):
        for q in self.p:
            q.kill()
        for q in self.p:
            q.wait()


def _unusual_records(kind, exe, g, tmp_path, repeats, spinners):
    from fastquick_amd import synth
    files = {}
    for mode in ("plain", "bgzf"):
        work = tmp_path / mode
        work.mkdir()
        paths = []
        for e, k in enumerate(("fq1", "fq2")):
            lines = open(g[k], "rb").read().split(b"\n")[:4 * 800]
            if kind == "long_names":
                for i in (100, 101, 300):
                    lines[4 * i] += b"_" + b"n" * (200 + i)
            elif kind == "tabs_in_names":
                for i in range(0, 800, 7):
                    lines[4 * i] += b"\tXX:Z:tag more"
            elif kind == "read_longer_than_the_rows" and e == 0:
                lines[4 * 400 + 1] *= 2; lines[4 * 400 + 3] *= 2
            elif kind == "read_of_one_base":
                lines[4 * 600 + 1] = lines[4 * 600 + 1][:1]; lines[4 * 600 + 3] = lines[4 * 600 + 3][:1]
            text = b"".join(x + b"\n" for x in lines)
            p = str(work / ("r%d.fq" % (e + 1) + (".gz" if mode == "bgzf" else "")))
            with open(p, "wb") as fh:
                fh.write(text if mode == "plain" else synth.bgzf_compress(text, threads=1, level=6, member=4000))
            paths.append(p)
        files[mode] = (work, paths)
    def run(mode):
        work, paths = files[mode]
        cmd = [exe, "align", "--index_prefix", g["prefix"][:-len(".FASTQuick.fa")], "--fastq_1", paths[0], "--fastq_2", paths[1], "--out_prefix", str(work / "o"),
               "--sam_out", "--batch_pairs", "256", "--chunk_pairs", "256"]
        return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    a = run("plain")
    refusal = kind.startswith("read_")
    assert a.returncode == (1 if refusal else 0)
    if refusal:
        # the records of every call before the refused one are printed (src/BwtMapper.cpp:2030-2092): two whole chunks of 256 pairs here
        assert len(a.stdout) > 100000
    with _Load(spinners):
        for _ in range(repeats):
            b = run("bgzf")
            assert b"front end on the device" in b.stderr or b.returncode
            assert b.returncode == a.returncode
            assert a.stdout == b.stdout, "SAM text of the device front end's run (%d bytes) is not the host reader's (%d)" % (len(b.stdout), len(a.stdout))
            if refusal:
                last = [r.stderr.decode(errors="replace").strip().splitlines()[-1].split("failed: ")[-1] for r in (a, b)]
                assert last[0] == last[1] and ("longer than the batch rows" in last[0] or "read length outside" in last[0])


@pytest.mark.parametrize("kind", ["long_names", "tabs_in_names", "read_longer_than_the_rows", "read_of_one_base"])
def test_cli_unusual_records_device_front_end_equals_the_host_reader(kind, golden_cases, emu_cli, tmp_path):
    """names of several hundred bytes, tab-separated fields behind the name, a read longer than --read_len allows, a read below the 15 bases the
    path takes: BGZF through the device front end == plain files through the host reader -- the same SAM text up to the same refusal.  A refusal
    is run several times beside busy-loop processes: the consumers of the last good call run on their own threads and must have written their
    records before the process ends (round 5: they were not waited for, and the output depended on the scheduler)."""
    _unusual_records(kind, emu_cli, golden_cases["qc"], tmp_path, repeats=8 if kind.startswith("read_") else 1, spinners=8 if kind.startswith("read_") else 0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["read_longer_than_the_rows", "read_of_one_base"])
def test_cli_refusal_keeps_the_records_before_it_on_the_gpu(kind, golden_cases, tmp_path):
    """the same refusals through the product's command line on the device, beside busy-loop processes"""
    exe = os.path.join(os.path.dirname(HERE), "fastquick_amd", "bin", "FASTQuick_amd")
    _unusual_records(kind, exe, golden_cases["qc"], tmp_path, repeats=6, spinners=16)
