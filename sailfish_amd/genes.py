"""Gene-level aggregation of quant.sf -- host mirror of sailfish::utils::readTranscriptToGeneMap
(src/SailfishUtils.cpp:438-507), TranscriptGeneMap::geneName (include/TranscriptGeneMap.hpp:94-140),
aggregateEstimatesToGeneLevel (src/SailfishUtils.cpp:929-1037) and generateGeneLevelEstimates (:1039-1088), the
`--geneMap` post-processing step of `sailfish quant`.

Host-side text processing in the reference and here: it reads the PRINTED quant.sf (so the 6-significant-digit
values), a few hundred kilobytes; there is nothing for the device to do.  The arithmetic is restated literally,
including two quirks a drop-in must keep:
  * totalTPM accumulates the RUNNING gene sum (`totalTPM += expVals[tpmIdx]` after the add, :1004-1009), so the
    TPM-weighted gene lengths are weighted by tpm_i / (sum of prefix sums), not by tpm_i / sum;
  * a transcript absent from the map is looked up with lower_bound and no equality test (:94-99): it lands on the
    next name in sorted order, and is "its own gene" only past the last name.
Lines of the output are in first-appearance order of the genes (the reference iterates an unordered_map: any order).
The GTF form of the map (`--geneMap x.gtf`, transcriptGeneMapFromGTF, :322-436) goes through libgff's GffReader in the
reference -- a third-party library that is not in the reference tree (CMakeLists.txt:408-419 fetches it), so its record
handling is restated from its use there: every record that carries a transcript_id makes (or extends) a transcript;
the grouping key is gene_id, gene_name, or any attribute named by `agg_key`; transcripts are ordered by name, genes
numbered by first appearance in that order.  Parity unpinned for this reader (no reference vector exists for it)."""
import bisect
import os

from .writer import fmt_g

DENORM_MIN = 4.9406564584124654e-324


class TranscriptGeneMap:
    """readTranscriptToGeneMap (:438-507): `transcript gene` pairs, whitespace separated; names sorted."""

    def __init__(self, pairs):
        gene_id, gene_names, t2g_unordered, names = {}, [], [], []
        for t, g in pairs:
            if g not in gene_id:
                gene_id[g] = len(gene_names); gene_names.append(g)
            names.append(t); t2g_unordered.append(gene_id[g])
        order = sorted(range(len(names)), key=lambda i: names[i])         # std::sort on the names (stable enough: ties keep any order)
        self.transcript_names = [names[i] for i in order]
        self.t2g = [t2g_unordered[i] for i in order]
        self.gene_names = gene_names

    @classmethod
    def from_file(cls, path):
        toks = open(path).read().split()                                  # `ifile >> transcript >> gene` until it fails
        return cls(list(zip(toks[0::2], toks[1::2])))

    @classmethod
    def from_gtf(cls, path, agg_key="gene_id"):
        """transcriptGeneMapFromGTF (src/SailfishUtils.cpp:322-436).  GFF2/GTF records: 9 tab-separated columns, the last
        one `key "value"; key "value"; ...`; lines starting with # are comments."""
        first_key = {}                                                    # transcript_id -> value of the grouping key (first seen)
        for line in open(path):
            if not line.strip() or line.startswith("#"):
                continue
            cols = line.rstrip("\n").split("\t")
            if len(cols) < 9:
                continue
            attrs = {}
            for field in cols[8].split(";"):
                field = field.strip()
                if not field:
                    continue
                k, _, v = field.partition(" ")
                attrs.setdefault(k, v.strip().strip('"'))
            t = attrs.get("transcript_id")
            if not t:
                continue                                                  # gene records and the like: not a transcript (isTranscript())
            if t not in first_key or first_key[t] is None:
                first_key[t] = attrs.get(agg_key, first_key.get(t))
        obj = cls.__new__(cls)
        gene_id, obj.gene_names, obj.transcript_names, obj.t2g = {}, [], [], []
        for t in sorted(first_key):                                       # std::sort by strcmp on the transcript ids (:392-395)
            g = first_key[t] if first_key[t] is not None else ""
            if g not in gene_id:
                gene_id[g] = len(obj.gene_names); obj.gene_names.append(g)
            obj.transcript_names.append(t); obj.t2g.append(gene_id[g])
        return obj

    def num_transcripts(self): return len(self.transcript_names)
    def num_genes(self): return len(self.gene_names)

    def gene_name(self, transcript_name):
        i = bisect.bisect_left(self.transcript_names, transcript_name)   # findTranscriptID: lower_bound, no equality test
        return self.gene_names[self.t2g[i]] if i < len(self.transcript_names) else transcript_name


def aggregate_estimates_to_gene_level(tgm: TranscriptGeneMap, quant_path: str) -> str:
    """aggregateEstimatesToGeneLevel (:929-1037): writes <quant_path minus extension>.genes.sf, returns its path."""
    comments, gene_exps, header = [], {}, True
    for line in open(quant_path).read().split("\n"):
        if not line.strip():
            continue
        if line.lstrip()[0] == "#":
            comments.append(line)
        elif header:
            comments.append(line); header = False                         # the header line is kept as a comment (:970-973)
        else:
            toks = line.split()
            if len(toks) < 3:
                raise ValueError("Any expression line must contain at least 3 tokens")
            rec = (toks[0], float(int(toks[1])), float(toks[2]), [float(x) for x in toks[3:]])   # stoi, stod, stod...
            gene_exps.setdefault(tgm.gene_name(rec[0]), []).append(rec)
    out_path = os.path.splitext(quant_path)[0] + ".genes.sf"
    with open(out_path, "w") as out:
        for c in comments:
            out.write(c + "\n")
        for gene, recs in gene_exps.items():
            ne = len(recs[0][3])
            exp_vals = [0.0] * ne
            total_tpm = 0.0
            for _, _, _, vals in recs:
                for i in range(ne):
                    exp_vals[i] += vals[i]
                total_tpm += exp_vals[0]                                  # the running sum, as in the reference
            gene_len = gene_eff = 0.0
            if total_tpm > DENORM_MIN:
                for _, length, eff, vals in recs:
                    frac = vals[0] / total_tpm
                    gene_len += length * frac; gene_eff += eff * frac
            else:
                frac = 1.0 / len(recs)
                for _, length, eff, _ in recs:
                    gene_len += length * frac; gene_eff += eff * frac
            out.write("\t".join([gene, fmt_g(gene_len), fmt_g(gene_eff)] + [fmt_g(v) for v in exp_vals]) + "\n")
    return out_path


def generate_gene_level_estimates(gene_map_path: str, est_dir: str, agg_key: str = "gene_id") -> str:
    """generateGeneLevelEstimates (:1039-1088): a map whose extension is .gtf is read as GTF, anything else as the
    two-column format."""
    if os.path.splitext(gene_map_path)[1] == ".gtf":
        tgm = TranscriptGeneMap.from_gtf(gene_map_path, agg_key)
    else:
        tgm = TranscriptGeneMap.from_file(gene_map_path)
    est = os.path.join(est_dir, "quant.sf")
    if not os.path.exists(est):
        raise ValueError(f"Attempting to compute gene-level esimtates, but could not \nfind isoform-level file {est}")
    return aggregate_estimates_to_gene_level(tgm, est)
