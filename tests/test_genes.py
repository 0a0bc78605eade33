"""Gene-level aggregation (`--geneMap`, src/SailfishUtils.cpp:929-1088): hand-computed expectations, quirks included."""
import os


def test_gene_level_aggregation(tmp_path):
    from sailfish_amd import genes
    (tmp_path / "map.tsv").write_text("tB g1\ntA g1\ntC g2\ntE g3\n")
    tgm = genes.TranscriptGeneMap.from_file(str(tmp_path / "map.tsv"))
    assert tgm.num_transcripts() == 4 and tgm.num_genes() == 3 and tgm.transcript_names == ["tA", "tB", "tC", "tE"]
    assert tgm.gene_name("tB") == "g1" and tgm.gene_name("tC") == "g2"
    assert tgm.gene_name("tD") == "g3"          # absent: lower_bound lands on tE (the reference's lookup has no equality test)
    assert tgm.gene_name("tZ") == "tZ"          # past the end: its own gene
    (tmp_path / "quant.sf").write_text("# sailfish (quasi) v0.10.0\nName\tLength\tEffectiveLength\tTPM\tNumReads\n"
                                       "tA\t1000\t800\t30\t60\ntB\t2000\t1800\t10\t40\ntC\t500\t300\t0\t0\ntZ\t700\t500\t5\t7\n")
    out = genes.generate_gene_level_estimates(str(tmp_path / "map.tsv"), str(tmp_path))
    assert os.path.basename(out) == "quant.genes.sf"
    lines = open(out).read().split("\n")
    assert lines[0] == "# sailfish (quasi) v0.10.0" and lines[1] == "Name\tLength\tEffectiveLength\tTPM\tNumReads"
    # g1: sums 40 / 100; totalTPM = 30 + (30 + 10) = 70 (running sum); length = 1000*30/70 + 2000*10/70 = 714.286
    assert lines[2] == "g1\t714.286\t600\t40\t100"
    # g2: not expressed: plain mean of the lengths
    assert lines[3] == "g2\t500\t300\t0\t0"
    # tZ is its own gene: totalTPM = 5, weights 1
    assert lines[4] == "tZ\t700\t500\t5\t7" and lines[5] == ""


def test_gene_map_from_gtf(tmp_path):
    """--geneMap x.gtf (transcriptGeneMapFromGTF, src/SailfishUtils.cpp:322-436): transcripts are the distinct
    transcript_id values (transcript, exon and CDS records alike), ordered by name; the key is gene_id, gene_name or a
    caller-named attribute"""
    from sailfish_amd import genes
    gtf = "\n".join([
        "##description: toy",
        'chr1\tsrc\tgene\t1\t900\t.\t+\t.\tgene_id "G1"; gene_name "alpha";',
        'chr1\tsrc\ttranscript\t1\t900\t.\t+\t.\tgene_id "G1"; transcript_id "tB"; gene_name "alpha"; tag "x";',
        'chr1\tsrc\texon\t1\t300\t.\t+\t.\tgene_id "G1"; transcript_id "tB"; gene_name "alpha";',
        'chr1\tsrc\texon\t1\t200\t.\t+\t.\tgene_id "G1"; transcript_id "tA"; gene_name "alpha"; tag "y";',      # no transcript record: made from its exon
        'chr2\tsrc\ttranscript\t5\t700\t.\t-\t.\tgene_id "G2"; transcript_id "tC"; gene_name "beta"; tag "x";',
        'chr2\tsrc\tCDS\t5\t100\t.\t-\t0\tgene_id "G2"; transcript_id "tC"; gene_name "beta";',
        ""])
    (tmp_path / "map.gtf").write_text(gtf)
    tgm = genes.TranscriptGeneMap.from_gtf(str(tmp_path / "map.gtf"))
    assert tgm.transcript_names == ["tA", "tB", "tC"] and tgm.gene_names == ["G1", "G2"] and tgm.t2g == [0, 0, 1]
    assert genes.TranscriptGeneMap.from_gtf(str(tmp_path / "map.gtf"), "gene_name").gene_names == ["alpha", "beta"]
    by_tag = genes.TranscriptGeneMap.from_gtf(str(tmp_path / "map.gtf"), "tag")
    assert by_tag.gene_names == ["y", "x"] and by_tag.t2g == [0, 1, 1]           # genes numbered by first appearance in transcript order
    (tmp_path / "quant.sf").write_text("Name\tLength\tEffectiveLength\tTPM\tNumReads\ntA\t1000\t800\t30\t60\ntB\t2000\t1800\t10\t40\ntC\t500\t300\t2\t3\n")
    out = genes.generate_gene_level_estimates(str(tmp_path / "map.gtf"), str(tmp_path))      # the .gtf extension selects the reader (:1050-1053)
    lines = open(out).read().split("\n")
    assert lines[1] == "G1\t714.286\t600\t40\t100" and lines[2] == "G2\t500\t300\t2\t3"


def _tgm_vectors():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_tgm_vectors.json")))["cases"]


def test_gene_lookup_matches_the_reference_header():
    """genes.TranscriptGeneMap.gene_name against TranscriptGeneMap::geneName of the reference's own header
    (include/TranscriptGeneMap.hpp:94-135, compiled unmodified into oracle/_ref/libtgm_ref.so; vectors committed by
    tests/golden/make_ref_tgm_vectors.py): names in the map, absent names that land on the next entry, names past the end"""
    from sailfish_amd import genes
    n = 0
    for case in _tgm_vectors():
        tgm = genes.TranscriptGeneMap.__new__(genes.TranscriptGeneMap)
        tgm.transcript_names, tgm.gene_names, tgm.t2g = case["transcripts"], case["genes"], case["t2g"]
        for q, want in zip(case["queries"], case["gene_of_query"]):
            assert tgm.gene_name(q) == want, (q, want)
            n += 1
    assert n > 300


def test_reference_header_still_gives_the_committed_vectors():
    """where oracle/_ref was built (this container only -- it is not shipped to the GPU box): the live library reproduces the fixture"""
    import ctypes as C
    import pytest
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libtgm_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libtgm_ref.so not built (no /root/reference here)")
    L = C.CDLL(so)
    L.ref_tgm_gene_names.restype = C.c_size_t
    L.ref_tgm_gene_names.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(C.c_size_t),
                                     C.POINTER(C.c_char_p), C.c_size_t, C.c_char_p, C.c_size_t]
    arr = lambda strs: (C.c_char_p * len(strs))(*[s.encode() for s in strs])
    for case in _tgm_vectors():
        out = C.create_string_buffer(1 << 20)
        L.ref_tgm_gene_names(arr(case["transcripts"]), len(case["transcripts"]), arr(case["genes"]), len(case["genes"]),
                             (C.c_size_t * len(case["t2g"]))(*case["t2g"]), arr(case["queries"]), len(case["queries"]), out, len(out))
        assert out.value.decode().split("\n") == case["gene_of_query"]
