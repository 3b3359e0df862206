"""hipGraph capture of every pooling head (forward + backward), INTEGRATION.md's claim made checkable:
    python tools/graph_rows.py [--quick]      # prints one JSON list (also gpurun_out/r5_graph_rows.json)
For each head at its BASELINE config shape: an eager run (the reference result), a capture with torch.cuda.graph
(forward, loss, backward - every C-ABI launch, the Newton-Schulz fork / join onto the helper queues included), three
replays, each compared BIT FOR BIT with the eager tensors, and the time of eager vs replay (HIP events, back to back).
tests/test_gpu_graph.py asserts on the same rows; bench.py attaches them to gpurun_out/bench_detail.json.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import hawkeye_amd.functional as F

dev = torch.device('cuda:0')
QUICK = '--quick' in sys.argv


def R(*shape, seed=0, relu=False, scale=1.0):
    t = torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale
    return (torch.relu(t) if relu else t).to(dev)


def head_bcnn(B=64):
    x = R(B, 512, 14, 14, seed=1, relu=True).requires_grad_(True)
    w, b = R(200, 512 * 512, seed=2, scale=0.01).requires_grad_(True), R(200, seed=3, scale=0.1).requires_grad_(True)
    y = torch.randint(0, 200, (B,), generator=torch.Generator().manual_seed(4)).to(dev)
    return [x, w, b], lambda: torch.nn.functional.cross_entropy(F.bilinear_pool_linear(x, w, b), y)


def head_cbcnn(B=16):
    plan = F.CbpPlan(*F.sketch_hashes(512, 512, 6000), 6000, dev)
    x = R(B, 512, 14, 14, seed=5, relu=True).requires_grad_(True)
    w, b = R(200, 6000, seed=6, scale=0.05).requires_grad_(True), R(200, seed=7, scale=0.1).requires_grad_(True)
    y = torch.randint(0, 200, (B,), generator=torch.Generator().manual_seed(8)).to(dev)
    return [x, w, b], lambda: torch.nn.functional.cross_entropy(F.linear(F.compact_bilinear_pool(x, plan), w, b), y)


def head_mpn(B=64):
    x = R(B, 256, 14, 14, seed=9, relu=True).requires_grad_(True)
    w, b = R(200, 32896, seed=10, scale=0.02).requires_grad_(True), R(200, seed=11, scale=0.1).requires_grad_(True)
    y = torch.randint(0, 200, (B,), generator=torch.Generator().manual_seed(12)).to(dev)

    def fn():                                     # MPNCOV.py:85-92 + MPN.forward :36-37; two-queue Newton-Schulz chain inside
        v = F.sqrtm_triuvec(F.covpool(x), 5, symmetric=True)
        return torch.nn.functional.cross_entropy(F.linear(v.view(B, -1), w, b), y)
    return [x, w, b], fn


def head_apcnn(B=16):
    feats = [R(B, 256, s, s, seed=13 + i).requires_grad_(True) for i, s in enumerate((56, 28, 14))]
    masks = [torch.sigmoid(R(B, 1, s, s, seed=16 + i) * 2.0) for i, s in enumerate((56, 28, 14))]
    x2 = R(B, 512, 56, 56, seed=19).requires_grad_(True)
    wp = [R(3, B, 256, seed=20), R(3, B, 256, seed=21)]
    wy = R(B, 512, 56, 56, seed=22)
    levels = ((8, 64, 5), (16, 128, 3), (32, 256, 1))

    def fn():                                     # APCNN.py:236-268 (pooled form), :444-476 (three levels, one launch), :478-531
        gap, sgap = F.att_pool_levels(feats, masks)
        tables = F.att_roi_select_levels(masks, levels, 448, 448, 8142, 0.05)
        box, drop = F.roi_boxes(tables, None, 8)
        yc = F.roi_crop_resize(x2, box, drop, False)
        return (gap * wp[0]).sum() + (sgap * wp[1]).sum() + (yc * wy).sum() * 1e-3
    return feats + [x2], fn


def head_osme(N=10):
    x = R(N, 2048, 7, 7, seed=23, relu=True).requires_grad_(True)
    w1, w2 = R(128, 2048, seed=24, scale=0.02).requires_grad_(True), R(2048, 128, seed=25, scale=0.05).requires_grad_(True)
    fc, fb = R(1024, 2048 * 49, seed=26, scale=0.003).requires_grad_(True), R(1024, seed=27, scale=0.1).requires_grad_(True)
    wt = R(N, 1024, seed=28)

    def fn():                                     # OSME.py:19-24,36-44 (one gate)
        z = F.osme_gap(x)
        m = torch.sigmoid(torch.relu(z @ w1.t()) @ w2.t()).unsqueeze(0)
        s = F.osme_scale(x, m)[0]
        return (F.linear(s.reshape(N, -1), fc, fb) * wt).sum()
    return [x, w1, w2, fc, fb], fn


def head_cin(B=20, hw=7):
    x = R(B, 2048, hw * hw, seed=29, relu=True, scale=0.5).requires_grad_(True)
    wt = torch.rand(B, generator=torch.Generator().manual_seed(30)).to(dev).requires_grad_(True)
    g1, g2 = R(B, 2048, hw * hw, seed=31), R(B, 2048, hw * hw, seed=32)

    def fn():                                     # CIN.py:31-34 (SCI), :51-54 (CCI): both branches and the gradient that reaches W_SCI
        y, w = F.cin_sci(x)
        return ((y * g1).sum() + (F.cin_cci(w, x, wt) * g2).sum()) * 1e-3
    return [x, wt], fn


HEADS = {'BCNN (pool + classifier, B=64)': head_bcnn, 'CBCNN (compact pool + classifier, B=16)': head_cbcnn,
         'MPN (covariance + Newton-Schulz + triuvec + classifier, B=64)': head_mpn,
         'APCNN (attention pooling x 3 + ROI select + crop / resize, B=16)': head_apcnn,
         'OSME (squeeze, gate, scale, part FC, N=10)': head_osme,
         'CIN (self- and contrastive channel interaction, B=20, 7x7 maps)': head_cin,
         'CIN (the same at 14x14 maps: stored-score forward, cin_ax_kernel products)': lambda: head_cin(20, 14)}


def time_loop(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run_head(name, build, iters=20):
    leaves, fn = build()

    def step():
        for t in leaves:
            t.grad = None
        loss = fn()
        loss.backward()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                          # warm-up off the default stream (torch's capture recipe)
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    loss = step()
    torch.cuda.synchronize()
    ref = [loss.detach().clone()] + [t.grad.detach().clone() for t in leaves]
    del loss      # (an eager loss kept alive keeps its AccumulateGrad nodes - bound to the default stream - alive: the captured
                  #  backward would then synchronise with the default stream inside the capture, and end-capture crashes)
    us_eager = time_loop(step, iters)
    for t in leaves:
        t.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gl = step()
    same, worst = True, 0.0
    for _ in range(3):
        for t in leaves:
            if t.grad is not None:
                t.grad.fill_(float('nan'))             # a replay that skipped a kernel would leave this behind
        g.replay()
        torch.cuda.synchronize()
        got = [gl.detach()] + [t.grad for t in leaves]
        for a, b in zip(got, ref):
            if not torch.equal(a, b):
                same = False
                worst = max(worst, float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)))
    us_replay = time_loop(g.replay, iters)
    return {'head': name, 'bit_identical_to_eager': same, 'worst_rel_diff': worst, 'us_eager': round(us_eager, 1),
            'us_graph_replay': round(us_replay, 1), 'replays_checked': 3}


if __name__ == '__main__':
    only = [a[7:] for a in sys.argv[1:] if a.startswith('--head=')]
    if only:                                        # child: one head
        name = list(HEADS)[int(only[0])]
        print('ROW ' + json.dumps(run_head(name, HEADS[name], iters=5 if QUICK else 20)), flush=True)
        sys.exit(0)
    # parent: every head in a process of its own (a capture that fails must not take the others with it)
    import subprocess
    rows = []
    for idx, name in enumerate(HEADS):
        cmd = [sys.executable, '-X', 'faulthandler', os.path.abspath(__file__), '--head=%d' % idx] + (['--quick'] if QUICK else [])
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
        got = [ln for ln in p.stdout.splitlines() if ln.startswith('ROW ')]
        if p.returncode == 0 and got:
            rows.append(json.loads(got[-1][4:]))
        else:
            rows.append({'head': name, 'error': f'rc={p.returncode}', 'stderr_tail': p.stderr[-1500:]})
    print(json.dumps(rows))
    if os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
        with open(os.path.join(ROOT, 'gpurun_out', 'r5_graph_rows.json'), 'w') as f:
            json.dump(rows, f, indent=1)
