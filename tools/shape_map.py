#!/usr/bin/env python
"""Per-kernel times and roofline fractions over a map of ASR-sized shapes off BASELINE.json's five points.

Two steps (the second needs no GPU):

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o map -- python tools/shape_map.py run > OUT/manifest.json
    python tools/shape_map.py report OUT/map_kernel_trace.csv OUT/manifest.json > profiles/r04_shape_map.md

`run` walks the shapes; for each it runs three entry points a few times -- log_softmax + rnnt_loss(gather=True)
(the reference's protocol, benchmark.py:62-70), the fused rnnt_loss_from_logits, and (ragged lengths)
rnnt_loss(compact=True) -- separated by marker fills whose grid size is unique, and prints a manifest.  `report`
cuts the kernel trace at the markers and prices every kernel with SURVEY.md 8(d)'s algorithmic bytes; the gathers -- two
dwords out of every 4V-byte row -- are priced a second time on the 128-byte lines those dwords live in (counted exactly in
`run`), the unit the memory system moves.
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK = 8000.0     # GB/s
REPS = 6
MARK0 = 7_000_001     # numel of marker i = MARK0 + 4096 * i (a fill kernel nobody else launches at that size)

# (N, T, U, V): T x U at V=128, N=32; the vocabulary axis; the batch axis
SHAPES = [(32, t, u, 128) for t in (250, 500, 1000) for u in (50, 100, 200)] + \
         [(32, 500, 100, 1024), (32, 500, 100, 4096), (16, 500, 100, 128), (64, 500, 100, 128), (64, 1000, 200, 128)]


def lengths(torch, N, T, U, ragged, seed):
    if not ragged:
        return torch.full((N,), T, dtype=torch.int32), torch.full((N,), U - 1, dtype=torch.int32)
    g = torch.Generator().manual_seed(seed)
    xn = torch.randint(T // 2, T + 1, (N,), generator=g, dtype=torch.int32)
    yn = torch.randint(U // 2, U, (N,), generator=g, dtype=torch.int32)
    xn += T - xn.max()
    yn += (U - 1) - yn.max()
    return xn, yn


def gather_line_bytes(torch, row_index, labels_of_cell, V):
    """128-byte lines holding the blank (column 0) or the label logit of the rows `row_index` (row r starts at byte
    4 V r), plus the 8 bytes written per cell."""
    off = row_index.to(torch.int64) * (V * 4)
    lines = torch.unique(torch.cat([off // 128, (off + labels_of_cell.to(torch.int64) * 4) // 128])).numel()
    return lines * 128.0 + 8.0 * row_index.numel()


def run():
    import torch
    import warp_rnnt
    import warp_rnnt_amd  # noqa: F401
    from warp_rnnt_amd import debug
    from warp_rnnt_amd import ops
    from warp_rnnt_amd.fused import rnnt_loss_from_logits
    dev = torch.device("cuda:0")
    sections = []
    nmark = [0]

    def mark():
        i = nmark[0]
        nmark[0] += 1
        torch.empty((MARK0 + 4096 * i,), dtype=torch.float32, device=dev).fill_(0.0)
        return i

    for (N, T, U, V) in SHAPES:
        g = torch.Generator(device=dev).manual_seed(N + T + U + V)
        xs = torch.randn((N, T, U, V), device=dev, generator=g)
        ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device=dev, generator=g)
        for ragged in (False, True):
            xn, yn = (t.to(dev) for t in lengths(torch, N, T, U, ragged, T + U))
            cells = int((xn.long() * (yn.long() + 1)).sum().item())
            lab = torch.zeros((N, U), dtype=torch.int64, device=dev)
            lab[:, :U - 1] = ys
            padded_lines = gather_line_bytes(torch, torch.arange(N * T * U, device=dev),
                                             lab[:, None, :].expand(N, T, U).reshape(-1), V)
            common = {"N": N, "T": T, "U": U, "V": V, "ragged": ragged, "live_cells": cells, "padded_cells": N * T * U,
                      "gather_line_bytes": padded_lines}

            def section(path, fn):
                fn()                                   # warm (allocator, lazy init)
                torch.cuda.synchronize()
                i = mark()
                for _ in range(REPS):
                    fn()
                j = mark()
                torch.cuda.synchronize()
                sections.append(dict(common, path=path, marker=i, marker_end=j, reps=REPS,
                                     lattice=debug.last_lattice_kernel()))

            section("log_softmax + rnnt_loss(gather=True)",
                    lambda: warp_rnnt.rnnt_loss(ops.log_softmax(xs), ys, xn, yn, gather=True))
            section("rnnt_loss_from_logits (fused)", lambda: rnnt_loss_from_logits(xs, ys, xn, yn))
            if ragged:
                lp = ops.log_softmax(xs)
                rows = torch.cat([lp[n, :int(xn[n]), :int(yn[n]) + 1].reshape(-1, V) for n in range(N)]).contiguous()
                labs = torch.cat([ys[n, :int(yn[n])] for n in range(N)]).contiguous()
                del lp
                clab = torch.cat([torch.cat([ys[n, :int(yn[n])].long(), torch.zeros(1, dtype=torch.int64, device=dev)])
                                  .repeat(int(xn[n])) for n in range(N)])
                common["gather_line_bytes"] = gather_line_bytes(torch, torch.arange(rows.shape[0], device=dev), clab, V)
                section("rnnt_loss(compact=True, bounds)",
                        lambda: warp_rnnt.rnnt_loss(rows, labs, xn, yn, compact=True, max_frames=T, max_labels=U - 1))
                del rows, labs
        del xs
        torch.cuda.empty_cache()
    print(json.dumps({"sections": sections, "mark0": MARK0}))


def algorithmic_bytes(name, s):
    """SURVEY.md 8(d) per-cell prices x the cells the launch covers (padded planes for the padded entry points)."""
    V, cells = s["V"], s["padded_cells"]
    if "compact" in s["path"]:
        cells = s["live_cells"]
    if "k_lsm" in name and "gather" in s["path"].lower() and "from_logits" not in s["path"]:
        return 8.0 * V * cells, "8V B/cell"
    if "k_lsm" in name:
        return (4.0 * V + 8) * cells, "4V+8 B/cell"
    if "k_to_diagonal" in name or "k_gather_compact" in name:
        return 16.0 * cells, "16 B/cell"
    if "k_lattice" in name:
        return 24.0 * cells, "24 B/cell"
    if "k_grads" in name:
        return 28.0 * cells, "28 B/cell"
    if "k_expand" in name:
        return (4.0 * V + 8) * cells, "4V+8 B/cell"
    return None, ""


def short(name):
    name = name.replace("void ", "").replace("rnnt::", "")
    return name.split("(")[0][:64]


def report(trace_csv, manifest_json):
    man = json.load(open(manifest_json))
    rows = list(csv.DictReader(open(trace_csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    grid_key = "Grid_Size" if "Grid_Size" in rows[0] else "Grid_Size_X"
    # marker i is a fill of MARK0 + 4096 i floats: four per thread, 256 threads per block -> a grid nobody else launches
    base = (MARK0 // 4 + 255) // 256 * 256 if MARK0 % 4 == 0 else ((MARK0 + 3) // 4 + 255) // 256 * 256
    at = {}
    for i, r in enumerate(rows):
        if "FillFunctor<float>" in r["Kernel_Name"]:
            gsz = int(r[grid_key])
            if gsz >= base and (gsz - base) % 1024 == 0:
                at.setdefault((gsz - base) // 1024, i)
    secs = man["sections"]
    print("# Shape map (round 4): per-kernel time and roofline fraction off BASELINE.json's five points\n")
    print("MI355X, one `rocprofv3 --kernel-trace` pass of `tools/shape_map.py run`; average over "
          f"{REPS} calls; algorithmic bytes per SURVEY.md 8(d) (padded cells for the padded entry points, live cells "
          "for the compact one); fraction of the 8 TB/s spec.  The lattice kernels are latency-bound: their fraction "
          "is shown for completeness.\n")
    worst = []
    for s in secs:
        a, b = at[s["marker"]] + 1, at[s["marker_end"]]
        ks = {}
        for r in rows[a:b]:
            nm = short(r["Kernel_Name"])
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            ks.setdefault(nm, []).append(d)
        span = (int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 / s["reps"] if b > a else 0.0
        print(f"## N={s['N']} T={s['T']} U={s['U']} V={s['V']} {'ragged' if s['ragged'] else 'full'} -- {s['path']}"
              f"  ({span:.1f} us per call, lattice: {s['lattice']})\n")
        print("| kernel | launches/call | avg us | algorithmic bytes | GB/s | of 8 TB/s | 128-byte lines touched | of 8 TB/s |")
        print("|---|---|---|---|---|---|---|---|")
        for nm, ds in sorted(ks.items(), key=lambda kv: -sum(kv[1])):
            avg = sum(ds) / len(ds)
            ab, rule = algorithmic_bytes(nm, s)
            if ab:
                gbs = ab / (avg * 1e-6) / 1e9
                frac, lines = gbs / HBM_PEAK, ""
                if ("k_to_diagonal" in nm or "k_gather_compact" in nm) and s.get("gather_line_bytes"):
                    lb = s["gather_line_bytes"]
                    frac = lb / (avg * 1e-6) / 1e9 / HBM_PEAK
                    lines = f"{lb / 1e6:.1f} MB | {frac:.3f}"
                else:
                    lines = " | "
                print(f"| `{nm}` | {len(ds) / s['reps']:.1f} | {avg:.1f} | {ab / 1e6:.1f} MB ({rule}) | {gbs:.0f} | {gbs / HBM_PEAK:.3f} | {lines} |")
                if avg > 20 and "k_lattice" not in nm and "prepare" not in nm:
                    worst.append((frac, nm, s, avg))
            else:
                print(f"| `{nm}` | {len(ds) / s['reps']:.1f} | {avg:.1f} | | | | | |")
        print()
    worst.sort(key=lambda w: w[0])
    print("## Streaming kernels furthest below the HBM roofline (launches longer than 20 us; the gathers on their lines)\n")
    print("| fraction | kernel | shape | path | avg us |")
    print("|---|---|---|---|---|")
    for fr, nm, s, avg in worst[:15]:
        print(f"| {fr:.3f} | `{nm}` | N={s['N']} T={s['T']} U={s['U']} V={s['V']} {'ragged' if s['ragged'] else 'full'} | {s['path']} | {avg:.1f} |")


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) >= 4 and sys.argv[1] == "report":
        report(sys.argv[2], sys.argv[3])
    else:
        sys.exit(__doc__)
