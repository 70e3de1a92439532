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


def _cells_slow(cells, gi, n_alt):
    """Presence and allele counts of one record, cell by cell (any ploidy, any FORMAT)."""
    pres = np.zeros(len(cells), dtype=bool)
    counts = np.zeros(n_alt + 1, dtype=np.int64)
    for j, cell in enumerate(cells):
        al = [-1 if a == "." else int(a) for a in cell.split(":")[gi].replace("|", "/").split("/")]
        for a in al:
            if 0 <= a <= n_alt:
                counts[a] += 1
        called = len(al) > 1 and all(a >= 0 for a in al)      # (haploid = padded with a missing allele: not called)
        if called and any(a != al[0] for a in al[1:]):
            pres[j] = True                                      # het
        elif called and al[0] > 0:
            pres[j] = True                                      # hom-alt
    return pres, counts


def _cells_fast(rest, n_samp, n_alt):
    """The same for the layout large call sets come in -- FORMAT is just GT and every cell is `a|b` / `a/b` with
    one-character alleles (0-9 or .), i.e. the sample columns are n_samp cells of width 3: one reshape instead of a loop
    over the cells.  None when the record is not of that shape (the caller then walks the cells)."""
    if len(rest) != 4 * n_samp - 1:
        return None
    a = np.frombuffer(rest + b"\t", dtype=np.uint8).reshape(n_samp, 4)
    sep_ok = ((a[:, 1] == 0x7C) | (a[:, 1] == 0x2F)).all() and (a[:, 3] == 0x09).all()
    al = a[:, (0, 2)].astype(np.int16) - 48                     # '0'..'9' -> 0..9, '.' -> -2
    digit = (al >= 0) & (al <= 9)
    if not sep_ok or not (digit | (al == -2)).all():
        return None
    a0, a1 = al[:, 0], al[:, 1]
    called = digit[:, 0] & digit[:, 1]
    pres = called & ((a0 != a1) | (a0 > 0))
    counted = al[digit & (al <= n_alt)]
    counts = np.bincount(counted, minlength=n_alt + 1).astype(np.int64)
    return pres, counts


def read_vcf(path, fast=True):
    """-> {'GT': uint8 (n, ceil(S/8)) numpy.packbits rows, 'AF': float64 (n, 1), 'samples': str (S,)}"""
    opener = gzip.open if path.endswith(".gz") else open
    samples = None
    rows, afs = [], []
    with opener(path, "rb") as fh:
        for raw in fh:
            if raw.startswith(b"##"):
                continue
            raw = raw.rstrip(b"\r\n")
            if raw.startswith(b"#CHROM"):
                samples = np.array(raw.decode().split("\t")[9:], dtype=str)
                continue
            if not raw:
                continue
            if samples is None:
                raise ValueError(f"{path}: a record in front of the #CHROM header line")
            f = raw.split(b"\t", 9)
            if len(f) < 10:
                raise ValueError(f"{path}: a record with fewer than ten columns")
            n_alt = len(f[4].split(b","))
            got = _cells_fast(f[9], len(samples), n_alt) if fast and f[8] == b"GT" else None
            if got is None:
                cells = f[9].decode().split("\t")
                if len(cells) != len(samples):
                    raise ValueError(f"{path}: a record with {len(cells)} sample columns, the header names {len(samples)}")
                got = _cells_slow(cells, f[8].decode().split(":").index("GT"), n_alt)
            pres, counts = got
            total = counts.sum()
            afs.append((counts[1:] / total).max() if total and n_alt else 0.0)
            rows.append(np.packbits(pres))
    if samples is None:
        raise ValueError(f"{path}: no #CHROM header line")
    width = (len(samples) + 7) // 8
    gt = np.array(rows, dtype=np.uint8).reshape(len(rows), width)
    return {"GT": gt, "AF": np.array(afs, dtype=np.float64).reshape(-1, 1), "samples": samples}
