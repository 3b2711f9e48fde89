"""Randomised check of the dense launches on segment-padded row lists (rgnn_pad_list_by_segment + rgnn_linear_fwd with
a1_panel_segment): random widths, ragged segments (empty ones, segments without list entries), with / without the second
operand block, statistics and bounds, against float64 torch.    python tools/fuzz_segment_linear.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radargnn_amd import ops


def one(rng, case):
    n_seg = int(rng.integers(1, 9))
    sizes = [int(rng.choice([0, 1, 37, 255, 256, 257, 700, 1500, 3000])) for _ in range(n_seg)]
    if sum(sizes) < 4096:
        sizes[int(rng.integers(0, n_seg))] += 4096                      # (enough rows for the matrix-pipe path)
    seg = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n_rows = int(seg[-1])
    k1 = int(rng.integers(1, 33)) * 16
    k2 = int(rng.choice([0, 0, 16, 64, 464]))
    n = int(rng.integers(9, 121)) * 4
    frac = float(rng.choice([0.0, 0.05, 0.5, 0.97, 1.0]))
    keep = rng.random(n_rows) < frac
    if rng.random() < 0.3 and n_seg > 1:
        f = int(rng.integers(0, n_seg)); keep[seg[f]:seg[f + 1]] = False   # a segment without list entries
    ids = np.nonzero(keep)[0].astype(np.int32)
    buf = np.full(n_rows, 987654, dtype=np.int32); buf[:len(ids)] = ids
    dev = torch.device("cuda")
    x = torch.from_numpy(rng.standard_normal((n_rows, k1)).astype(np.float32) * 3).to(dev)
    a2 = torch.from_numpy(rng.standard_normal((n_rows, k2)).astype(np.float32)).to(dev) if k2 else None
    w = torch.from_numpy((rng.standard_normal((n, k1 + k2)) / np.sqrt(k1 + k2)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(dev)
    table = torch.from_numpy(np.stack([np.stack([rng.uniform(-1, 1, k1), rng.uniform(0.2, 2.0, k1), rng.uniform(-1, 1, k1)]) for _ in range(n_seg)]).astype(np.float32)).to(dev)
    use_tab = rng.random() < 0.8
    relu = bool(rng.random() < 0.5)
    if os.environ.get("FUZZ_VERBOSE"):
        print(f"  segs {sizes} k1 {k1} k2 {k2} n {n} listed {len(ids)} tab {use_tab} relu {relu}", flush=True)
    lst, total, tiles, start = ops.pad_list_by_segment(torch.from_numpy(buf).to(dev), torch.tensor([len(ids)], device=dev), torch.from_numpy(seg).to(dev))
    out = torch.full((n_rows, n), -7.0, device=dev)
    stats = torch.zeros((max(ops.stat_panels(lst.numel()), 1), ops.STAT_ROWS, n), device=dev)
    with ops.bound_tracking(dev):
        if rng.random() < 0.7:                                           # (f16x2 form; otherwise bf16x3)
            ops.set_bound(x, ops.make_bound(x.abs().max()))
            ops.set_bound(table, ops.make_bound(torch.tensor(3 * 4.5 * 2.0 + 1.0, device=dev) if False else (x.abs().max() * 2.0 + 1.0)))
            if a2 is not None:
                ops.set_bound(a2, ops.make_bound(a2.abs().max()))
        try:
            ops.linear(x, w, b, a2=a2, out=out, relu=relu, row_index=lst, m_dev=total, stats_out=stats,
                       a1_affine=table if use_tab else None, a1_affine_tiles=tiles if use_tab else None, padded_row_list=True)
        except ops.RgnnError as e:
            return "skipped: " + str(e)[:60]
    torch.cuda.synchronize()
    frame_of = np.searchsorted(seg, np.arange(n_rows), side="right") - 1
    xa = x.double().cpu()
    if use_tab:
        xa = torch.relu(ops.apply_table_reference(x.cpu(), table.cpu()[frame_of]))
    full = torch.cat([xa, a2.double().cpu()], dim=1) if a2 is not None else xa
    ref = full @ w.double().cpu().t() + b.double().cpu()
    if relu:
        ref = torch.relu(ref)
    got = out.double().cpu()
    listed = np.zeros(n_rows, dtype=bool); listed[ids] = True
    err = 0.0
    if listed.any():
        err = float((got[listed] - ref[listed]).abs().max() / ref[listed].abs().max().clamp_min(1e-30))
    untouched = bool((got[~listed] == -7.0).all())
    sp = start.cpu().numpy()
    serr = 0.0
    for f in range(n_seg):
        rows = ids[(ids >= seg[f]) & (ids < seg[f + 1])]
        s1 = ops.stats_to_sums(stats[sp[f]:sp[f + 1]])[1].cpu().numpy()
        r1 = ref[rows].numpy().sum(0) if len(rows) else np.zeros(n)
        serr = max(serr, float(np.abs(s1 - r1).max() / max(1.0, np.abs(r1).max())))
    ok = err <= 4e-6 and untouched and serr <= 1e-4 and int(total) == sum((int(((ids >= seg[f]) & (ids < seg[f + 1])).sum()) + 255) // 256 * 256 for f in range(n_seg))
    return ("ok " if ok else "FAIL ") + f"case {case}: segs {sizes} k1 {k1} k2 {k2} n {n} listed {len(ids)} tab {use_tab} relu {relu} err {err:.2e} stats {serr:.2e} untouched {untouched}"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for c in range(cases):
        if os.environ.get("FUZZ_VERBOSE"):
            print("case", c, flush=True)
        r = one(rng, c)
        if not r.startswith("ok"):
            print(r); bad += r.startswith("FAIL")
    print(f"{cases} cases, {bad} failures")


if __name__ == "__main__":
    main()
