"""Minimal text VCF genotype reader (build-owned; used to cross-check fixtures).

Restates what ``utmos/convert.py:43-88`` extracts through scikit-allel (absent here):
presence = het or hom-alt call, AF = max alt-allele frequency over called alleles.
scikit-allel's handling of half-missing / haploid calls is third-party code not in
the reference tree, so anything beyond fully-called diploid GTs is *parity unpinned*.
"""
import gzip

import numpy as np


def read_vcf_text(path):
    op = gzip.open if path.endswith(".gz") else open
    samples = None
    rows = []
    afs = []
    with op(path, "rt") as fh:
        for line in fh:
            if line.startswith("##"):
                continue
            f = line.rstrip("\n").split("\t")
            if line.startswith("#CHROM"):
                samples = np.array(f[9:], dtype=str)
                continue
            fmt = f[8].split(":")
            gi = fmt.index("GT")
            n_alt = len(f[4].split(","))
            pres = np.zeros(len(f) - 9, dtype=bool)
            counts = np.zeros(n_alt + 1, dtype=np.int64)
            for j, cell in enumerate(f[9:]):
                gt = cell.split(":")[gi].replace("|", "/").split("/")
                al = [-1 if a == "." else int(a) for a in gt]
                for a in al:
                    if a >= 0:
                        counts[a] += 1
                called = all(a >= 0 for a in al)
                if len(al) > 1 and called and any(a != al[0] for a in al[1:]):
                    pres[j] = True          # het (allel GenotypeArray.is_het)
                elif called and al[0] > 0 and all(a == al[0] for a in al):
                    pres[j] = True          # hom-alt (is_hom_alt)
            an = counts.sum()
            with np.errstate(invalid="ignore", divide="ignore"):
                freq = counts / an
            afs.append(freq[1:].max() if n_alt else np.nan)
            rows.append(pres)
    gt = np.array(rows, dtype=bool).reshape(len(rows), len(samples))
    return {"GT": np.packbits(gt, axis=1), "AF": np.array(afs, dtype=np.float64).reshape(-1, 1),
            "samples": samples}
