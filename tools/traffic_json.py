#!/usr/bin/env python3
"""profiles/*_traffic.json from the summary of tools/profile_pmc.sh's PMC passes (FETCH_SIZE, WRITE_SIZE per dispatch of gfw_jit_kernel; KiB units), stamped with the
identity of the kernel source inside the library that was profiled (abi.kernel_source_id): bench.py quotes `roofline.traffic` only for a library that matches.
usage: tools/traffic_json.py <profile dir with summary.txt> <out.json> [frames per dispatch = 10]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gyroflow_amd import abi  # noqa: E402


def main():
    d, out = sys.argv[1], sys.argv[2]
    fpl = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
    txt = open(os.path.join(d, "summary.txt")).read()
    val = {}
    for name in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
        m = re.search(r"%s\s+mean/dispatch = ([0-9.e+]+)" % name, txt)
        if m:
            val[name] = float(m.group(1))
    k = re.search(r"gfw_jit_kernel\s+calls=(\d+) total_ns=(\d+) avg_ns=([0-9.]+)", txt)
    j = {"workload": "C2 3840x2160 YUV422P16LE bilinear, clip launches of %g frames of gfw_jit_kernel" % fpl,
         "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum, separate passes (tools/profile_pmc.sh); summary: %s" % os.path.join(d, "summary.txt"),
         "kernel_source_id": abi.kernel_source_id(),
         "fetch_size_kib_per_frame": val["FETCH_SIZE"] / fpl, "fetch_correction": 2.0, "write_size_kib_per_frame": val["WRITE_SIZE"] / fpl,
         "tcc_hit_rate": val.get("TCC_HIT_sum", 0.0) / max(val.get("TCC_HIT_sum", 0.0) + val.get("TCC_MISS_sum", 0.0), 1.0),
         "kernel_avg_ns_per_dispatch": float(k.group(3)) if k else None,
         "note": "gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes (MI355X_MICROARCH.md, HBM): doubled.",
         "calibration": "profiles/r06_counter_calibration.txt: kernels that move 1 GiB once report FETCH_SIZE x 2.000 = bytes for dwordx4 AND dword reads, WRITE_SIZE x 1.000 = bytes "
                        "for dwordx4, dword and 2-byte stores (tools/fetch_calib.hip, same box, separate --pmc passes): the factors hold for this library's access widths"}
    j["hbm_bytes_per_frame"] = int((j["fetch_size_kib_per_frame"] * 2.0 + j["write_size_kib_per_frame"]) * 1024)
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j))


if __name__ == "__main__":
    main()
