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
