#!/usr/bin/env python3
"""Per-kernel energy of the C4 pass (profiles/r6_energy.json).

The pass is bound by the 1,400 W package cap, so what a kernel costs is joules, not stall cycles.  For each of
the step's kernels the whole step is captured as a HIP graph with THAT launch issued R times (the kernel sees exactly the data of
a real step: inputs produced by the launches in front of it; the similarity GEMM's list header and counts are re-zeroed between
repeats, two one-block torch kernels), every graph is replayed back to back for --seconds, and
    E(kernel) = (E(graph with R launches) - E(plain step)) / (R - 1),   t(kernel) likewise,
from the package's accumulated-energy counter (rocm-smi --showenergycounter) read before and after the loop, with the mean of the
sampled package power x wall time beside it as a cross-check, and the held shader clock.
usage: python tools/energy_table.py [--workload c4_40kx10k] [--seconds 3] [--repeat 9] [--out profiles/r6_energy.json]"""
import argparse, json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def smi(*flags):
    try:
        return subprocess.run(['rocm-smi'] + list(flags), capture_output=True, text=True, timeout=10).stdout
    except Exception as e:  # noqa: BLE001
        return ''


def energy_uj():
    m = re.search(r'Accumulated Energy \(uJ\): *([0-9.eE+]+)', smi('--showenergycounter'))
    return float(m.group(1)) if m else None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = threading.Event()
        self.w, self.mhz = [], []

    def run(self):
        while not self.stop.is_set():
            t = smi('--showclocks', '--showpower')
            m = re.findall(r'sclk clock level:? *\d*:? *\((\d+)Mhz\)', t)
            p = re.findall(r'Package Power \(W\): *([0-9.]+)', t)
            if m:
                self.mhz.append(int(m[0]))
            if p:
                self.w.append(float(p[0]))


class Repeat:
    """ops.profiler hook: the launches named `name` are issued `n` times (ops._call)."""
    def __init__(self, name, n):
        self.name, self.n, self.enabled, self.seen = name, n, True, 0

    def begin(self, name):
        pass

    def end(self, name):
        pass

    def repeat(self, name):
        if name == self.name:
            self.seen += 1
            return self.n
        return 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='c4_40kx10k')
    ap.add_argument('--seconds', type=float, default=3.0)
    ap.add_argument('--repeat', type=int, default=9)
    ap.add_argument('--precision', default='fp16')
    ap.add_argument('--no-scores', action='store_true', help='count-only mode (S not materialised)')
    ap.add_argument('--out', default='profiles/r6_energy.json')
    args = ap.parse_args()
    import laff_amd.model.model as M
    from laff_amd import ops, synth
    from laff_amd.dist import HipBackend, evaluate_sharded
    M.FC_PRECISION = 'fp16x3'
    dev = torch.device('cuda:0')
    Nt, Nv, heads, d, frames = synth.WORKLOADS[args.workload]
    spec = synth.SPECS.get(args.workload)
    model = synth.build_model(heads, d, dev, frames=frames, seed=1237, spec=spec)
    vis, txt, gt, _ = synth.make_features(Nt, Nv, dev, frames=frames, seed=1237, spec=spec)
    be = HipBackend(model, args.precision)
    pin = torch.zeros(8, dtype=torch.float64).pin_memory()
    want_scores = not args.no_scores

    def step():
        return evaluate_sharded(be, vis, txt, gt, Nt, Nv, heads, metrics_out=pin, want_scores=want_scores)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ref = tuple(pin[:7].tolist())

    orig_gemm = ops.sim_gemm_banded
    R = args.repeat

    def gemm_repeated(st, *a, **k):
        S = None
        for i in range(R):
            if i:
                st.pairs[:4].zero_()
                st.count.zero_()
            S = orig_gemm(st, *a, **k)
        return S

    def capture(target):
        ops.profiler = None
        ops.sim_gemm_banded = orig_gemm
        if target == 'sim_gemm':
            ops.sim_gemm_banded = gemm_repeated
        elif target is not None:
            ops.profiler = Repeat(target, R)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                step()
        finally:
            seen = ops.profiler.seen if ops.profiler is not None else (1 if target == 'sim_gemm' else 0)
            ops.profiler = None
            ops.sim_gemm_banded = orig_gemm
        g.replay()
        torch.cuda.synchronize()
        return g, seen

    def run(g, seconds):
        # settle the package at its cap first
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:
            for _ in range(50):
                g.replay()
            torch.cuda.synchronize()
        s = Sampler()
        e0 = energy_uj()
        s.start()
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(50):
                g.replay()
            n += 50
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
        e1 = energy_uj()
        s.stop.set()
        s.join()
        w = s.w[1:] if len(s.w) > 2 else s.w
        mw = sum(w) / max(len(w), 1)
        return {'replays': n, 'ms_per_replay': 1e3 * el / n, 'mean_power_w': mw, 'power_samples': len(w),
                'mhz': (sum(s.mhz[1:]) / max(len(s.mhz[1:]), 1)) if len(s.mhz) > 1 else None,
                'j_per_replay_counter': ((e1 - e0) * 1e-6 / n) if (e0 is not None and e1 is not None) else None,
                'j_per_replay_power_x_time': mw * el / n}

    targets = [None, 'fc_act_bn', 'fuse', 'sim_gemm', 'rank_resolve', 'rank_metrics']
    rows = {}
    for tg in targets:
        g, seen = capture(tg)
        r = run(g, args.seconds)
        r['launches_repeated_per_step'] = seen
        rows[tg or 'step'] = r
        print(tg or 'step', json.dumps(r), file=sys.stderr, flush=True)
        del g
    # metrics unchanged by the repeats (the last repeat leaves the step's own result)
    step(); torch.cuda.synchronize()
    assert tuple(pin[:7].tolist()) == ref
    base = rows['step']
    key = 'j_per_replay_counter' if base['j_per_replay_counter'] is not None else 'j_per_replay_power_x_time'
    K = heads * d
    # algorithmic work per launch (SURVEY section 8d / DESIGN section 4)
    nfeat_rows = 4 * (Nt + Nv)
    work = {'fc_act_bn': ('flop', 2.0 * nfeat_rows * 512 * K), 'fuse': ('byte', 4.0 * (Nt + Nv) * K * 5),
            'sim_gemm': ('flop', 2.0 * Nt * Nv * K), 'rank_resolve': ('pair', None), 'rank_metrics': ('rank', float(Nt))}
    table = {}
    for tg in targets[1:]:
        r = rows[tg]
        nl = max(r['launches_repeated_per_step'], 1) * (R - 1)       # extra launches per replay
        de, dt = r[key] - base[key], r['ms_per_replay'] - base['ms_per_replay']
        ent = {'extra_launches_per_replay': nl, 'ms': dt / (R - 1), 'joule': de / (R - 1),
               'watt_while_running': (de / (dt * 1e-3)) if dt > 0 else None, 'mhz_in_loop': r['mhz'],
               'note': 'per step: the sum over the %d launch(es) of this name' % max(r['launches_repeated_per_step'], 1)}
        kind, amount = work[tg]
        if amount:
            ent['algorithmic_' + kind] = amount
            ent['pJ_per_algorithmic_' + kind] = 1e12 * ent['joule'] / amount
        table[tg] = ent
    out = {'workload': args.workload, 'scores': want_scores, 'repeat': R, 'seconds_per_loop': args.seconds, 'energy_source': key,
           'step': base, 'kernels': table,
           'sum_of_kernels_joule': sum(v['joule'] for v in table.values()), 'sum_of_kernels_ms': sum(v['ms'] for v in table.values()),
           'loops': rows}
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    json.dump(out, open(args.out, 'w'), indent=1)
    print(json.dumps({k: out[k] for k in ('workload', 'scores', 'energy_source', 'sum_of_kernels_joule', 'sum_of_kernels_ms')}))
    for k, v in table.items():
        print('%-13s %.4f ms  %.4f J  %s W  %s' % (k, v['ms'], v['joule'], '%.0f' % v['watt_while_running'] if v['watt_while_running'] else '-',
                                                  ' '.join('%s %.3f' % (kk, vv) for kk, vv in v.items() if kk.startswith('pJ'))))
    print('step          %.4f ms  %.4f J  %.0f W  %s MHz' % (base['ms_per_replay'], base[key], base['mean_power_w'], base['mhz']))


if __name__ == '__main__':
    main()
