"""Minimal text VCF genotype reader for `utmos select file.vcf[.gz]` inputs.

The reference reads VCFs through scikit-allel (utmos/convert.py:43-88), a third-party library that
is not part of the reference tree: presence = het or hom-alt call, AF = max alt-allele frequency over
called alleles.  This reader reproduces that for fully called diploid genotypes (its GT bytes equal
the reference's chunk0/chunk1 fixtures).  A haploid call ("1", chrX/chrY males) is NOT counted as present:
scikit-allel pads it to a diploid call with a missing second allele, which is neither het nor hom-alt.
Half-missing and haploid calls are *parity unpinned*: no fixture of the reference holds one.  So is AF on
multi-allelic rows: this reader follows utmos/convert.py:75-77 (largest alt-allele frequency), while the
reference's own chunk0.jl / chunk1.jl fixtures carry the FIRST alt allele's frequency on their five multi-allelic
rows (an older convert, presumably) -- and no golden of the reference scores a VCF input with --af.
"""
import gzip

import numpy as np


def read_vcf(path):
    """-> {'GT': uint8 (n, ceil(S/8)) numpy.packbits rows, 'AF': float64 (n, 1), 'samples': str (S,)}"""
    opener = gzip.open if path.endswith(".gz") else open
    samples = None
    rows, afs = [], []
    with opener(path, "rt") as fh:
        for line in fh:
            if line.startswith("##"):
                continue
            f = line.rstrip("\n").split("\t")
            if line.startswith("#CHROM"):
                samples = np.array(f[9:], dtype=str)
                continue
            gi = f[8].split(":").index("GT")
            n_alt = len(f[4].split(","))
            pres = np.zeros(len(f) - 9, dtype=bool)
            counts = np.zeros(n_alt + 1, dtype=np.int64)
            for j, cell in enumerate(f[9:]):
                al = [-1 if a == "." else int(a) for a in cell.split(":")[gi].replace("|", "/").split("/")]
                for a in al:
                    if 0 <= a <= n_alt:
                        counts[a] += 1
                called = len(al) > 1 and all(a >= 0 for a in al)      # (haploid = padded with a missing allele: not called)
                if called and any(a != al[0] for a in al[1:]):
                    pres[j] = True                                      # het
                elif called and al[0] > 0:
                    pres[j] = True                                      # hom-alt
            total = counts.sum()
            afs.append((counts[1:] / total).max() if total and n_alt else 0.0)
            rows.append(pres)
    if samples is None:
        raise ValueError(f"{path}: no #CHROM header line")
    gt = np.array(rows, dtype=bool).reshape(len(rows), len(samples))
    return {"GT": np.packbits(gt, axis=1), "AF": np.array(afs, dtype=np.float64).reshape(-1, 1), "samples": samples}
