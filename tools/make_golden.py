#!/usr/bin/env python3
"""Re-encode the reference's test fixtures as build-owned data files under tests/golden/.

Run once in the build container (needs /root/reference); the outputs are committed and
are the only thing that travels to the GPU box.  Data only: packed genotype bytes, AF
values, sample names, the reference's expected TSVs and its three small option files.

  * chunk{0,1,2}.jl  -> chunk{0,1,2}.npz   via tools/jl_static.py (opcode walk, no unpickling)
  * chunk_tiny.vcf   -> tiny.npz           via utmos_amd/vcfio.py (text parse)
  * answer_key/*.txt -> answer_key/*.txt   the goldens the reference's suite actually uses
                                           (repo_utils/utmos_ssshtests.sh:81-235)
  * weights.txt / subset.txt / exclude.txt copied (data)

Cross-check performed here: the GT bytes parsed from chunk{0,1}.vcf.gz text equal the GT
bytes found in chunk{0,1}.jl.
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
from jl_static import read_jl                              # noqa: E402
from utmos_amd.vcfio import read_vcf as read_vcf_text      # noqa: E402  (the build's own text VCF reader)

REF = "/root/reference/repo_utils"
OUT = os.path.join(HERE, "..", "tests", "golden")

# case -> (inputs, argv after the inputs, golden file, value mode)   [utmos_ssshtests.sh line]
CASES = {
    "select_first":          (["chunk2"], [], "select_first.txt"),                                   # :81 (chunk2.vcf missing -> chunk2.jl)
    "select_intcnt":         (["chunk1"], ["--count", "10"], "select_intcnt.txt"),                   # :87
    "select_floatcnt":       (["chunk2"], ["--count", "0.01"], "select_floatcnt.txt"),               # :93
    "select_fileout":        (["chunk1"], [], "select_fileout.txt"),                                 # :99
    "select_multi":          (["chunk0", "chunk2"], [], "select_multi.txt"),                         # :105-121
    "select_exclude":        (["chunk0", "chunk1"], ["-c", "20", "--exclude", "NA21117"], "select_exclude.txt"),   # :123
    "select_weights":        (["chunk0"], ["-c", "20", "--weights", "weights.txt"], "select_weights.txt"),         # :129
    "select_af":             (["chunk0", "chunk1"], ["-c", "20", "--af"], "select_af.txt"),          # :135
    "select_weightsaf":      (["chunk0", "chunk1"], ["-c", "5", "--af", "--weights", "weights.txt"], "select_weightsaf.txt"),  # :141
    "select_tiny":           (["tiny"], ["-c", "20"], "select_tiny.txt"),                            # :148
    "select_one_af":         (["chunk1"], ["-c", "0.005", "--af"], "select_one_af.txt"),             # :154
    "select_weights_subset": (["chunk0"], ["--subset", "subset.txt", "-c", "5", "--weights", "weights.txt"], "select_weights_subset.txt"),  # :161
    "select_af_subset":      (["chunk0"], ["--subset", "subset.txt", "-c", "5", "--af"], "select_af_subset.txt"),  # :168
    # hdf5 path (:218-235): values stored as float32 (select.py:218-223) -> --af-dtype f32 in the build
    "select_af_h5":          (["chunk0", "chunk1"], ["-c", "20", "--af", "--af-dtype", "f32"], "select_af_h5.txt"),
}


def main():
    os.makedirs(os.path.join(OUT, "answer_key"), exist_ok=True)
    for c in ("chunk0", "chunk1", "chunk2"):
        d = read_jl(f"{REF}/test_files/{c}.jl")
        assert d["GT"].dtype == np.uint8 and d["AF"].dtype == np.float64
        if os.path.exists(f"{REF}/test_files/{c}.vcf.gz"):
            v = read_vcf_text(f"{REF}/test_files/{c}.vcf.gz")
            assert (v["GT"] == d["GT"]).all() and (v["samples"] == d["samples"]).all(), c
        np.savez_compressed(os.path.join(OUT, f"{c}.npz"), GT=d["GT"], AF=d["AF"].reshape(-1),
                            samples=np.asarray(d["samples"], dtype="U"))
        print(c, d["GT"].shape, d["AF"].shape, len(d["samples"]))
    t = read_vcf_text(f"{REF}/test_files/chunk_tiny.vcf")
    np.savez_compressed(os.path.join(OUT, "tiny.npz"), GT=t["GT"], AF=t["AF"].reshape(-1),
                        samples=np.asarray(t["samples"], dtype="U"))
    print("tiny", t["GT"].shape, len(t["samples"]))
    for name in ("weights.txt", "subset.txt", "exclude.txt"):
        shutil.copyfile(f"{REF}/test_files/{name}", os.path.join(OUT, name))
    for case, (_, _, gold) in CASES.items():
        shutil.copyfile(f"{REF}/answer_key/{gold}", os.path.join(OUT, "answer_key", gold))
    for p in (os.path.join(OUT, n) for n in os.listdir(OUT)):
        if os.path.isfile(p):
            os.chmod(p, 0o644)
    for n in os.listdir(os.path.join(OUT, "answer_key")):
        os.chmod(os.path.join(OUT, "answer_key", n), 0o644)
    with open(os.path.join(OUT, "cases.json"), "w") as fh:
        json.dump({k: {"inputs": v[0], "args": v[1], "golden": v[2]} for k, v in CASES.items()}, fh, indent=1)


if __name__ == "__main__":
    main()
