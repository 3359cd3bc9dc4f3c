"""The CLI's FASTQ tokenizer (fq_cli.cpp, after kseq_read3_fpc) against the REAL reference's reader on oddly formatted input.

CPU tier, build container only: the front end is linked over the host-loop library (tests/emu, test infrastructure) and the
expected SAM text comes from oracle/_ref/fq_ref_driver, which tokenises with the reference's own kseq / bwa_read_seq_with_hash_dev.
Skipped where the reference build is absent (the GPU box runs the committed goldens instead)."""
import gzip
import os
import subprocess

import pytest

import oracle_binding as ob

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
pytestmark = pytest.mark.skipif(not os.path.exists(ob.REF_DRIVER), reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def emu_cli():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libfq_emu.so", "FASTQuick_emu"])
    return os.path.join(EMU_DIR, "FASTQuick_emu")


def records(path):
    with open(path, "rb") as fh:
        lines = fh.read().split(b"\n")
    return [(lines[i][1:], lines[i + 1], lines[i + 3]) for i in range(0, len(lines) - 3, 4)]


def wrap(s, w):
    return b"\n".join(s[i:i + w] for i in range(0, len(s), w))


def crlf(end, i, nm, s, q):
    return b"@" + nm + b"\r\n" + s + b"\r\n+\r\n" + q + b"\r\n"


def multiline(end, i, nm, s, q):            # sequence and quality wrapped at different widths, name repeated after '+'
    return b"@" + nm + b"\n" + wrap(s, 60) + b"\n+" + nm + b"\n" + wrap(q, 50) + b"\n"


def comments(end, i, nm, s, q):             # comment after a tab / blank, /1 /2 suffixes, lower-case and IUPAC / '.' bases
    sep = (b"\t", b" ")[i & 1]
    t = bytearray(s.lower() if i % 3 == 0 else s)
    if i % 5 == 0:
        t[7] = ord("R")
    if i % 7 == 0:
        t[11] = ord(".")
    return b"@" + nm + b"/" + (b"1", b"2")[end] + sep + b"1:N:0:ACGT extra\n" + bytes(t) + b"\n+\n" + q + b"\n"


def blank_tail(end, i, nm, s, q):
    return b"@" + nm + b"\n" + s + b"\n+\n" + q + b"\n"


def multiline_seq(end, i, nm, s, q):        # wrapped sequence, quality in one piece beginning with '@' or '+'
    q = (b"@", b"+")[i & 1] + q[1:]
    return b"@" + nm + b"\n" + wrap(s, 70) + b"\n+\n" + q + b"\n"


def ragged(end, i, nm, s, q):               # read lengths 100..150 (>= 96: the filter never looks past a read's end)
    n = 100 + (i * 7 + end * 13) % 51
    return b"@" + nm + b"\n" + s[:n] + b"\n+\n" + q[:n] + b"\n"


def short_mixed(end, i, nm, s, q):          # 15..150 bp: under 96 bp the filter also sees what the slot's earlier reads left (Q7)
    n = 15 + (i * 37 + (i // 60) * 29 + end * 13) % 136
    return b"@" + nm + b"\n" + s[:n] + b"\n+\n" + q[:n] + b"\n"


def long_names(end, i, nm, s, q):
    return b"@" + nm + b":" + b"x" * (20 + (i * 37) % 200) + b"\n" + s + b"\n+\n" + q + b"\n"


def mate_names(end, i, nm, s, q):           # the two files name their mates differently, with different lengths
    return b"@" + nm + (b".a", b".mate2")[end] + b"\n" + s + b"\n+\n" + q + b"\n"


VARIANTS = {
    "mate_names": (mate_names, b""),
    "multiline_seq": (multiline_seq, b""),
    "ragged": (ragged, b""),
    "short_mixed": (short_mixed, b""),
    "long_names": (long_names, b""),
    "second_file_short": (blank_tail, b"SHORT2"),
    "crlf": (crlf, b""),
    "multiline": (multiline, b""),
    "comments": (comments, b""),
    "blank_tail": (blank_tail, b"\n\n\n"),
    "no_final_newline": (blank_tail, None),
}


# container "bgzf": the same text as a BGZF file of small members -- the command line then inflates and tokenises on the device (fq_frontend_*;
# in this tier: the same kernel bodies in the host-loop library) and hands over to the host's reader where the text stops being four plain
# lines per record; the output must not depend on who read what
@pytest.mark.parametrize("container", ["gzip", "bgzf"])
@pytest.mark.parametrize("variant", list(VARIANTS))
def test_cli_tokenizer_matches_reference_reader(variant, container, golden_cases, emu_cli, tmp_path):
    g = golden_cases["repeat" if variant == "mate_names" else "basic"]     # "repeat" has pairs with one mate filtered
    fmt, tail = VARIANTS[variant]
    batch = 60 if variant == "short_mixed" else g["batch"]     # 7 batches: every slot is reused three times
    if variant == "mate_names":
        batch = 512                                            # one short batch: a full one would trip the reference's name check
    fq = []
    for end, key in enumerate(("fq1", "fq2")):
        body = b"".join(fmt(end, i, nm.split()[0], s, q) for i, (nm, s, q) in enumerate(records(g[key])))
        if tail == b"SHORT2":                # (the second file was meant to end 3 records early; what the reference does then is not defined --
            tail = b""                       #  its loop takes the first file's count for both -- and the driver refuses it: both files stay whole)
        body = body[:-1] if tail is None else body + tail
        path = str(tmp_path / ("reads_%d.fq.gz" % (end + 1)))
        if container == "bgzf":
            from fastquick_amd import synth
            with open(path, "wb") as fh:
                fh.write(synth.bgzf_compress(body, threads=2, level=6, member=777))
        else:
            with gzip.open(path, "wb") as fh:
                fh.write(body)
        fq.append(path)
    ref = subprocess.run([ob.REF_DRIVER, "align", g["prefix"], fq[0], fq[1], str(tmp_path / "ref_out"), "--batch", str(batch)],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    cmd = [emu_cli, "align", "--index_prefix", g["prefix"][:-len(".FASTQuick.fa")], "--fastq_1", fq[0], "--fastq_2", fq[1],
           "--out_prefix", str(tmp_path / "cli"), "--sam_out", "--batch_pairs", str(batch), "--chunk_pairs", str(batch)]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if ref.returncode != 0:      # input the reference's reader refuses (e.g. CR LF line ends) is refused here too, with its message
        assert run.returncode != 0, "the reference rejects this input: " + ref.stderr.decode(errors="replace")[-300:]
        assert ref.stderr.strip().split(b"\n")[-1].split(b",")[0] in run.stderr, (ref.stderr[-300:], run.stderr[-300:])
        return
    assert run.returncode == 0, run.stderr.decode(errors="replace")[-2000:]
    with open(str(tmp_path / "ref_out.sam"), "rb") as fh:
        want = fh.read()
    if variant in ("ragged", "short_mixed"):
        # The reference prints QUAL as a C string out of a slot buffer that is never terminated (src/BwtMapper.cpp:549-558): after a
        # longer read in the same slot the column carries that read's tail and is longer than SEQ (not valid SAM).  That column is
        # not modelled; everything else, alignments included, must agree.
        def cols(text):
            return [[c for k, c in enumerate(ln.split(b"\t")) if k != 10] for ln in text.split(b"\n")]
        assert cols(run.stdout) == cols(want)
        assert all(len(f[9]) == len(f[10]) for f in (ln.split(b"\t") for ln in run.stdout.split(b"\n")) if len(f) > 10)
        assert any(len(f[9]) != len(f[10]) for f in (ln.split(b"\t") for ln in want.split(b"\n")) if len(f) > 10)
        # ... and the command line says so, loudly; with --strict_reference it stops instead (the fence was silent)
        assert b"WARNING - reads of unequal lengths" in run.stderr
        strict = subprocess.run(cmd + ["--strict_reference"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert strict.returncode != 0 and b"--strict_reference: reads of unequal lengths" in strict.stderr
        return
    assert run.stdout == want
    assert b"WARNING - reads of unequal lengths" not in run.stderr, "reads of one length: nothing to warn about"
    if container == "bgzf":
        assert b"front end on the device" in run.stderr
        handed_over = b"read on by the host's reader" in run.stderr
        assert handed_over == (variant in ("multiline_seq", "multiline", "blank_tail", "no_final_newline")), run.stderr.decode(errors="replace")[-1500:]


def test_cli_refuses_a_read_len_that_would_expose_slot_reallocation(golden_cases, emu_cli, tmp_path):
    """The other fence (src/BwtMapper.cpp:536-546: a read longer than read_len makes the reference re-allocate its slot): invisible while
    read_len >= 96, the read filter's window -- the reference's own value is 151 -- and refused below that."""
    g = golden_cases["basic"]
    cmd = [emu_cli, "align", "--index_prefix", g["prefix"][:-len(".FASTQuick.fa")], "--fastq_1", g["fq1"], "--fastq_2", g["fq2"],
           "--out_prefix", str(tmp_path / "cli"), "--sam_out", "--read_len", "80"]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode != 0 and b"--read_len must be at least 96" in run.stderr


def test_cli_fq_list_runs_every_pair_as_its_own_stream(golden_cases, emu_cli, tmp_path):
    """--fq_list (src/BwtMapper.cpp:232-262): one header, then every listed pair mapped as an independent stream (own drand48 seed,
    insert-size history, read slots) -- the concatenation of the reference's outputs for the pairs run one by one."""
    g = golden_cases["repeat"]          # repeats: the drand48 stream decides placements, so a stream that was not re-seeded would show
    recs = [records(g[k]) for k in ("fq1", "fq2")]
    n = len(recs[0])
    parts, want = [], b""
    for k, (lo, hi) in enumerate(((0, n // 3), (n // 3, n))):
        fq = []
        for end in range(2):
            path = str(tmp_path / ("part%d_%d.fq" % (k, end + 1)))
            with open(path, "wb") as fh:
                fh.write(b"".join(blank_tail(end, i, nm, s, q) for i, (nm, s, q) in enumerate(recs[end][lo:hi])))
            fq.append(path)
        parts.append(fq)
        ob.run_reference(g["prefix"], fq[0], fq[1], str(tmp_path / ("ref%d" % k)), "--batch", 100)
        with open(str(tmp_path / ("ref%d.sam" % k)), "rb") as fh:
            text = fh.read()
        want += text if k == 0 else b"".join(ln + b"\n" for ln in text.split(b"\n") if ln and not ln.startswith(b"@"))
    lst = str(tmp_path / "fq.list")
    with open(lst, "w") as fh:
        fh.write("# comment line\n%s\t%s\n%s %s\n" % (parts[0][0], parts[0][1], parts[1][0], parts[1][1]))
    cmd = [emu_cli, "align", "--index_prefix", g["prefix"][:-len(".FASTQuick.fa")], "--fq_list", lst,
           "--out_prefix", str(tmp_path / "cli"), "--sam_out", "--batch_pairs", "100", "--chunk_pairs", "200"]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode == 0, run.stderr.decode(errors="replace")[-2000:]
    assert run.stdout == want
