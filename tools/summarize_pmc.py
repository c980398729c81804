#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

usage: summarize_pmc.py <fetch_results.db> <write_results.db> <out_prefix>

Units / corrections (MI355X_MICROARCH.md, section HBM): both counters are in KiB-of-64-byte-requests as the
gfx94x formulas define them (TCC_EA0_*REQ x 64 B / 1024); on gfx950 a wide coalesced read is tallied at half
its size, so FETCH bytes are DOUBLED here; WRITE_SIZE is reported as is (uncalibrated).  Infinity-Cache hits
are counted as traffic.  Values are averages per launch."""
import json
import sqlite3
import sys

CLASSES = [("gemm_wgrad", ("gemm_sym_kernel<true, true", "gemm_wgrad_group_kernel", "wgrad_wide_kernel")), ("gemm_dgrad", "gemm_sym_kernel<false, true"),
           ("gemm_fwd", ("gemm_sym_kernel<false, false", "gemm_ws_kernel")), ("gemm_ln", "gemm_ln_kernel"), ("row_chain_bwd", ("row_chain_bwd_kernel", "row_chain_bwd_pipe_kernel", "row_chain_bwd_split_kernel")),
           ("row_chain", ("row_chain_kernel", "row_chain_pipe_kernel", "row_chain_split_kernel")), ("wfrag_build", "wfrag_build_kernel"), ("gemm_lnbwd", "gemm_lnbwd_kernel"),
           ("attn_fwd", ("attn_fwd_kernel", "attn_fwd64_kernel", "attn_xs_fwd_kernel")), ("attn_bwd_dq", "attn_bwd_dq_kernel"),
           ("attn_bwd_dkv", "attn_bwd_dkv_kernel"), ("attn_bwd", ("attn_bwd_kernel", "attn_bwd64_kernel")), ("ln_bwd", "ln_bwd_kernel")]


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
                     "group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out, lines = {}, ["# per-launch HBM traffic (bytes); FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as reported",
                      "%-16s %8s %14s %14s %14s" % ("kernel class", "launches", "read_bytes", "write_bytes", "total_bytes")]
    for cls, pat in CLASSES:
        pats = pat if isinstance(pat, tuple) else (pat,)
        n = rd = wr = 0.0
        for name, (cnt, avg) in fetch.items():
            if any(q in name for q in pats):
                n += cnt
                rd += cnt * avg * 1024.0 * 2.0
        for name, (cnt, avg) in write.items():
            if any(q in name for q in pats):
                wr += cnt * avg * 1024.0
        if n:
            out[cls] = {"launches": int(n), "read_bytes": rd / n, "write_bytes": wr / n, "bytes": (rd + wr) / n}
            lines.append("%-16s %8d %14.0f %14.0f %14.0f" % (cls, n, rd / n, wr / n, (rd + wr) / n))
    with open(sys.argv[3] + ".txt", "w") as f:
        f.write("\n".join(lines) + "\n")
    import os
    out["git_sha"] = os.environ.get("GIT_SHA", "unrecorded")      # the GPU box has no .git: the caller passes the SHA
    with open(sys.argv[3] + ".json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
