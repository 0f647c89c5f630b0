#!/usr/bin/env python3
"""bench.py -- LAFF retrieval hot path on MI355X: text-video cosine pairs/s (+ R@1/5/10/MedR).

    python bench.py [--gpus N --steps K --warmup W --workload c4_40kx10k --precision fp16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full pass of the hot path over the synthetic workload with every input already resident in HBM:
8 FC projections (fp32 MFMA) -> 2 LAFF fusions -> operand packing -> text x video similarity GEMM writing the fp32
score matrix -> ground-truth rank counts -> R@K / MedR / mAP on the host.  Default workload: BASELINE.json's headline
shape, 40k texts x 10k videos x (4+4 features of 512-d), one head of d = 512 (`configs[3]`; it fits one GPU).
With N > 1 the SAME total problem is sharded by video rows (strong scaling) with one RCCL all-gather of the text
embeddings (laff_amd/dist.py: the loop being sharded is /root/reference/model/model.py:1057-1077).  Rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no launcher environment (no RANK / WORLD_SIZE): this process -- which has not touched the GPU -- starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD process, forwards rank 0's
JSON line to its own stdout and exits with the child's code.  LAFF_BENCH_DRYRUN=1 runs the same launch + collective path on CPU
(gloo, oracle-backed stand-in kernels from tests/dist_util.py, a toy problem): what the CPU test of the self-launch uses.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
L2_PEAK_GBS = 34500.0         # MI355X_MICROARCH.md: 8 x 4 MiB L2, ~34.5 TB/s aggregate
GRAPH_CHECK_REPLAYS = 64       # replays of the two captured steps that are checked against the eager step before the W warm-up replays
MFMA_PEAK_TFLOPS = {'f32': 157.3, 'f16': 2500.0, 'bf16': 2500.0}


class StageTimer:
    """HIP events on torch's current stream == the stream liblaff_hip launches on (ops._context binds it)."""

    def __init__(self):
        self.events = []      # (name, event) ; name None = step start
        self.enabled = True

    def start(self):
        if self.enabled:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.events.append((None, e))

    def mark(self, name):
        if self.enabled:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.events.append((name, e))

    def totals(self):
        """name -> (total ms, count) over all recorded steps."""
        out = {}
        prev = None
        for name, e in self.events:
            if name is not None and prev is not None:
                ms = prev.elapsed_time(e)
                t, c = out.get(name, (0.0, 0))
                out[name] = (t + ms, c + 1)
            prev = e
        return out


class LaunchProfiler:
    """Per-launch timing (ops.profiler hook).  Eager mode: HIP events on the stream the kernel is launched on.
    Stamp mode (inside a graph capture; this HIP runtime refuses event-record nodes there): a one-thread laff_stamp launch in front of and
    behind every C-ABI call writes the device's constant-rate wall clock into a slot buffer -- every replay of the captured graph
    re-writes the slots, so the differences are the durations of the kernels, and of the gaps between them, AS THE GRAPH RUNS THEM.
    A stamp costs one empty launch (calibrate(): the interval between two back-to-back stamps, subtracted once per measured launch)."""

    MAX_SLOTS = 512

    def __init__(self):
        self.spans = []
        self.enabled = False
        self.stamps = None          # int64 device tensor while in stamp mode
        self.names = []             # stamp mode: name of launch i (slots 2i, 2i + 1)
        self._open = None

    def begin(self, name):
        if not self.enabled:
            return
        if self.stamps is not None:
            from laff_amd import ops
            ops.stamp(self.stamps, 2 * len(self.names))
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self._open = e

    def end(self, name):
        if not self.enabled:
            return
        if self.stamps is not None:
            from laff_amd import ops
            ops.stamp(self.stamps, 2 * len(self.names) + 1)
            self.names.append(name)
            if 2 * len(self.names) + 2 > self.MAX_SLOTS:
                raise RuntimeError('LaunchProfiler: more than %d launches in one captured step' % (self.MAX_SLOTS // 2))
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.spans.append((name, self._open, e))

    def totals(self):
        out = {}
        for name, a, b in self.spans:
            t, c = out.get(name, (0.0, 0))
            out[name] = (t + a.elapsed_time(b), c + 1)
        return out


def cpu_baseline(workload, sample_nt, sample_nv, heads, d, seed, spec=None):
    """The oracle (numpy restatement of the reference, `oracle/`) timed on this host's cores on a bounded sample of the
    same workload, in the reference's own shape: batch-64 block loop with per-block re-normalisation
    (model/model.py:1057-1077), then argsort-free count ranks.  kind = "port"."""
    from laff_amd import synth
    from oracle import laff_oracle as O
    dev = torch.device('cuda:0')
    model = synth.build_model(heads, d, dev, seed=seed, spec=spec)
    vis, txt, gt, _ = synth.make_features(sample_nt, sample_nv, dev, seed=seed, spec=spec)
    vid_names = list(model.vis_net.opt.vid_feats)
    vnt, tnt = list(model.vis_net.opt.vis_no_transform), list(model.txt_net.opt.txt_no_transform)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    vis_np, txt_np, gt_np = synth.to_numpy_dict(vis), synth.to_numpy_dict(txt), gt.cpu().numpy()
    enc = {'rnn_encoder': 'rnn_encoding', 'bow_encoder': 'bow_encoding', 'w2v_encoder': 'w2v_encoding',
           'CLIP_encoder': 'CLIP_encoding'}
    names = list(model.txt_net.encoder_name_list)
    del model, vis, txt
    torch.cuda.empty_cache()
    bs = 64
    t0 = time.perf_counter()
    att_v = O.attention_from_sd(sd, 'vis_net.attention_layer.', heads, False, False)
    att_t = O.attention_from_sd(sd, 'txt_net.attention_layer.', heads, False, False)
    vb = []
    for s in range(0, sample_nv, bs):
        specs = [O.feature_spec(sd, 'vis_net.VisMutiTransformNet.%s.' % n, vis_np[n][s:s + bs], 'tanh', heads, n in vnt)
                 for n in vid_names]
        vb.append((np.arange(s, min(sample_nv, s + bs)), O.fuse_tower(specs, att_v, heads)))
    tb = []
    for s in range(0, sample_nt, bs):
        specs = [O.feature_spec(sd, 'txt_net.transform_layer.%s_transform.' % e, txt_np[enc[e]][s:s + bs], 'tanh', heads, e in tnt)
                 for e in names]
        tb.append((np.arange(s, min(sample_nt, s + bs)), O.fuse_tower(specs, att_t, heads)))
    S = O.predict_blocked(tb, vb, sample_nt, sample_nv)
    s_gt = S[np.arange(sample_nt), gt_np]
    ranks = (S > s_gt[:, None]).sum(axis=1) + 1
    metrics = O.eval_from_positions([[r] for r in ranks])
    dt = time.perf_counter() - t0
    try:        # threads the numpy BLAS actually runs (its pool is what does the work); the host may have more cores
        from threadpoolctl import threadpool_info
        cores = max([int(p.get('num_threads', 1)) for p in threadpool_info()] or [1])
    except Exception:  # noqa: BLE001
        cores = os.cpu_count()
    main = {'value': sample_nt * sample_nv / dt, 'unit': 'pairs/s', 'cores': cores, 'host_cores': os.cpu_count(), 'kind': 'port',
            'sample': '%dx%d slice of the same synthetic workload (seed, generator, weights), numpy oracle in the reference\'s '
                      'batch-64 block-loop shape (towers + per-block cosine), ranks by counting, %.1f s on the GPU box host '
                      '(%d BLAS threads of %d host cores)' % (sample_nt, sample_nv, dt, cores, os.cpu_count()),
            'r1': metrics[0]}
    # BASELINE.md section 3 variants.  (a) reference-shaped INCLUDING its ranking stage: full-matrix argsort + the per-query label loop
    # + evaluation.eval on the label matrix (predictor.py:232-246), on a row sample (the label matrix alone is 8 B per pair);
    # the score stage is charged pro rata.  (b) vectorised upper bound: whole-matrix towers, ONE GEMM, ranks by counting.
    variants = []
    try:
        na = min(sample_nt, 4000)
        t1 = time.perf_counter()
        inds = np.argsort(S[:na], axis=1)
        label = np.zeros((na, sample_nv))
        for i in range(na):
            ind = inds[i][::-1]
            label[i][np.where(ind == gt_np[i])[0]] = 1
        m_a = O.eval_label_matrix(label)
        dt_a = time.perf_counter() - t1 + dt * na / sample_nt
        variants.append({'kind': 'port-blockloop-argsort', 'value': na * sample_nv / dt_a, 'unit': 'pairs/s', 'cores': cores,
                         'sample': 'first %d texts x %d videos: block-loop scores (pro rata) + np.argsort + per-query label loop + '
                                   'eval on the label matrix, %.1f s' % (na, sample_nv, dt_a), 'r1': m_a[0]})
        del inds, label
        t2 = time.perf_counter()
        specs_v = [O.feature_spec(sd, 'vis_net.VisMutiTransformNet.%s.' % n, vis_np[n], 'tanh', heads, n in vnt) for n in vid_names]
        specs_t = [O.feature_spec(sd, 'txt_net.transform_layer.%s_transform.' % e, txt_np[enc[e]], 'tanh', heads, e in tnt) for e in names]
        ve_all, te_all = O.fuse_tower(specs_v, att_v, heads), O.fuse_tower(specs_t, att_t, heads)
        S2 = O.txt2vis_matrix_fast(te_all, ve_all)
        r2 = (S2 > S2[np.arange(sample_nt), gt_np][:, None]).sum(axis=1) + 1
        m_b = O.eval_from_positions([[r] for r in r2])
        dt_b = time.perf_counter() - t2
        variants.append({'kind': 'port-vectorised', 'value': sample_nt * sample_nv / dt_b, 'unit': 'pairs/s', 'cores': cores,
                         'sample': '%dx%d: whole-matrix towers, one GEMM over concatenated heads, ranks by counting, %.1f s'
                                   % (sample_nt, sample_nv, dt_b), 'r1': m_b[0]})
    except Exception as e:  # noqa: BLE001
        variants.append({'kind': 'error', 'sample': str(e)})
    main['variants'] = variants
    return main


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def visible_gpus():
    """GPU agents of this node, counted from the KFD topology in sysfs (a node with SIMDs is a GPU), cut down by HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES; torch.cuda.device_count() -- which on some ROCm builds reaches hipGetDeviceCount, i.e. initialises the
    runtime in this parent process -- only when sysfs has no topology."""
    n = 0
    root = '/sys/class/kfd/kfd/topology/nodes'
    try:
        for node in os.listdir(root):
            with open(os.path.join(root, node, 'properties')) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            if int(props.get('simd_count', '0')) > 0:
                n += 1
    except OSError:
        return torch.cuda.device_count()
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(',') if x.strip() != '']))
    return n


def self_launch(n, argv):
    """Parent of a `--gpus N` run without a launcher: N ranks under torch.distributed.run as a child process (never an exec: this
    image refuses to replace a process image once anything may have initialised the GPU), rank 0's line relayed."""
    import subprocess
    dry = os.environ.get('LAFF_BENCH_DRYRUN') == '1'
    if not dry:
        have = visible_gpus()                     # (from sysfs: the parent stays clear of the HIP runtime altogether)
        if have < n:
            print('bench.py: --gpus %d but this node has %d visible GPU(s)' % (n, have), file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL's peer mappings fail without it on this driver
    env.setdefault('OMP_NUM_THREADS', '8' if not dry else '2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout:
        if ln.lstrip().startswith('{') and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)                 # anything else a rank printed to fd 1
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print('bench.py: the ranks exited 0 without printing a JSON line', file=sys.stderr)
        rc = 1
    return rc


def dryrun_main(args):
    """LAFF_BENCH_DRYRUN=1: the launch / rendezvous / collective / timing / reporting path of an N-rank run on CPU -- gloo instead
    of RCCL, the oracle-backed stand-in of the per-rank kernels (tests/dist_util.py) on a toy problem.  Not a measurement."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from dist_util import OracleBackend, problem
    from laff_amd.dist import evaluate_sharded, evaluate_sharded_by_text, evaluate_sharded_v16, gathered_bytes, shard_bounds
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world > 1:
        dist.init_process_group('gloo')
    xt, xv, gt, Wt, Wv = problem()
    Nt, Nv = len(xt), len(xv)
    t0, t1 = shard_bounds(Nt, world, rank)
    v0, v1 = shard_bounds(Nv, world, rank)
    be = OracleBackend(Wt, Wv)
    vis_l, txt_l, gt_t = {'x': torch.from_numpy(xv[v0:v1])}, {'x': torch.from_numpy(xt[t0:t1])}, torch.from_numpy(gt)
    out = {}
    for kind, fn in (('video', evaluate_sharded), ('text', evaluate_sharded_by_text), ('video16', evaluate_sharded_v16)):
        for _ in range(max(1, args.warmup)):
            res = fn(be, vis_l, txt_l, gt_t, Nt, Nv, 1)
        if world > 1:
            dist.barrier()
        ts = time.perf_counter()
        for _ in range(args.steps):
            res = fn(be, vis_l, txt_l, gt_t, Nt, Nv, 1)
        if world > 1:
            dist.barrier()
        el = torch.tensor([time.perf_counter() - ts], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        out[kind] = (float(el.item()), res)
    if rank == 0:
        from laff_amd.dist import choose_sharding
        head = choose_sharding(Nt, Nv) if args.shard == 'auto' else args.shard
        others = [k for k in ('video', 'text', 'video16') if k != head]
        el, res = out[head]
        m = res['metrics']

        def gb(k):
            return gathered_bytes(k, Nt, Nv, Wt.shape[1], world, 4096) if k == 'video16' else gathered_bytes(k, Nt, Nv, Wt.shape[1], world)
        line = {'metric': 'text-video cosine pairs/sec', 'value': float(Nt) * Nv * args.steps / el, 'unit': 'pairs/s', 'n_gpus': world,
                'rccl_ranks': dist.get_world_size() if dist.is_initialized() else 1, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': 1e3 * el / args.steps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
                'dtype': 'dry run (numpy oracle stand-in)', 'data': 'synthetic',
                'config': {'workload': 'DRYRUN %d texts x %d videos (CPU, gloo): exercises the launcher, not the kernels' % (Nt, Nv),
                           'shard': head, 'shard_arg': args.shard, 'backend': 'gloo'},
                'alt_shard': {'shard': others[0], 'ms_per_step': 1e3 * out[others[0]][0] / args.steps,
                              'ranks_equal': bool(torch.equal(out[others[0]][1]['ranks'], res['ranks']))},
                'alt_shards': [{'shard': k, 'ms_per_step': 1e3 * out[k][0] / args.steps,
                                'ranks_equal': bool(torch.equal(out[k][1]['ranks'], res['ranks'])),
                                'gathered_bytes_per_step': gb(k)} for k in others],
                'gathered_bytes_per_step': gb(head),
                'quality': {'R@1': m[0], 'R@5': m[1], 'R@10': m[2], 'MedR': m[3], 'meanr': m[4], 'mir': m[5], 'mAP': m[6]},
                'collective_ms': None, 'roofline': None, 'cpu_baseline': None}
        os.write(json_fd, (json.dumps(line) + '\n').encode())
    if dist.is_initialized():
        dist.destroy_process_group()


def capture_graph(fn):
    """fn() captured as one HIP graph (thread-local capture mode), replayed once."""
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        fn()
    g.replay()
    torch.cuda.synchronize()
    return g


def timed_replays(graphs, steps):
    """`steps` replays alternating between the captures (step k + 1 is enqueued while step k finishes); wall seconds."""
    torch.cuda.synchronize()
    ts = time.perf_counter()
    ev = [torch.cuda.Event(), torch.cuda.Event()]
    for k in range(steps):
        graphs[k % len(graphs)].replay()
        ev[k % 2].record()
        if k:
            ev[(k - 1) % 2].synchronize()
    torch.cuda.synchronize()
    return time.perf_counter() - ts


_STAMP_COST_MS = {}


def stamp_cost_ms(dev):
    """Interval between two back-to-back laff_stamp launches inside a captured graph (median of 64): what one stamp adds to the
    interval it closes."""
    key = str(dev)
    if key not in _STAMP_COST_MS:
        from laff_amd import ops
        khz = ops.wall_clock_khz(dev)
        buf = torch.zeros(66, dtype=torch.int64, device=dev)
        ops.stamp(buf, 0)
        torch.cuda.synchronize()

        def chain():
            for i in range(65):
                ops.stamp(buf, i)
        g = capture_graph(chain)
        ds = []
        for _ in range(5):
            g.replay()
            torch.cuda.synchronize()
            t = buf[:65].cpu().numpy().astype(np.int64)
            ds.extend(np.diff(t).tolist())
        del g
        _STAMP_COST_MS[key] = (float(np.median(ds)) / khz, khz)
    return _STAMP_COST_MS[key]


def instrumented_replays(fn, prof, n):
    """The step captured once more with a laff_stamp launch in front of and behind every C-ABI launch (LaunchProfiler stamp mode), replayed
    n times: per entry point the mean ms per step and launches per step AS THE GRAPH RUNS THEM (one stamp's cost subtracted from every
    measured launch), the mean time between consecutive launches, and the mean first-stamp-to-last-stamp span.  None on failure."""
    dev = torch.device('cuda', torch.cuda.current_device())
    try:
        cost, khz = stamp_cost_ms(dev)
        prof.stamps = torch.zeros(prof.MAX_SLOTS, dtype=torch.int64, device=dev)
        prof.names = []
        prof.enabled = True
        try:
            g = capture_graph(fn)
        finally:
            prof.enabled = False
        names, buf = list(prof.names), prof.stamps
        if not names:
            return None
        per, gap_tot, span_tot = {}, 0.0, 0.0
        # every sample = the LAST of a run of back-to-back replays: a replay behind a host sync starts on an idle chip whose clocks are
        # still ramping (the package needs tens of milliseconds of load to settle at its cap), and its kernels came out up to 10 % slower
        # than the same kernels in the timed region (seen: sum of the stamped kernels 1.063 ms against ms_per_step 0.964)
        t_one = time.perf_counter()
        g.replay()
        torch.cuda.current_stream().synchronize()
        t_one = max(time.perf_counter() - t_one, 1e-5)
        # ~40 ms of continuous load in front of every sample (40 replays of a 1 ms step, 2 of a 30 ms one)
        warm = int(os.environ.get('LAFF_BENCH_STAMP_WARM', str(max(2, min(40, int(0.04 / t_one))))))
        for _ in range(n):
            for _w in range(warm):
                g.replay()
            g.replay()
            torch.cuda.current_stream().synchronize()
            t = buf[:2 * len(names)].cpu().numpy().astype(np.int64)
            for i, name in enumerate(names):
                ms = max(float(t[2 * i + 1] - t[2 * i]) / khz - cost, 0.0)
                a, c = per.get(name, (0.0, 0))
                per[name] = (a + ms, c + 1)
            gap_tot += sum(float(t[2 * i + 2] - t[2 * i + 1]) / khz for i in range(len(names) - 1))
            span_tot += float(t[2 * len(names) - 1] - t[0]) / khz
        del g
        return {'launches': {k: (a / n, c // n) for k, (a, c) in per.items()}, 'gaps_ms': gap_tot / n, 'span_ms': span_tot / n,
                'n_launches': len(names), 'stamp_cost_ms': cost}
    except Exception as e:  # noqa: BLE001
        print('warning: instrumented capture failed (%s)' % e, file=sys.stderr)
        return None
    finally:
        prof.stamps, prof.names, prof.enabled = None, [], False


def emulate_shard(backend, vis, txt, gt, Nt, Nv, heads, G, scheme, steps, warmup, pins, prof, profile_steps):
    """Rank 0's share of a G-rank pass of this workload on ONE device with no collectives (laff_amd.dist.EmulatedComm): its towers on
    Nt/G texts + Nv/G videos, then -- 'video': the operand of all Nt gathered text rows, the banded GEMM Nt x Nv/G, resolve -- or
    'text': the operand of all Nv gathered video rows, the banded GEMM Nt/G x Nv.  The peers' rows / exact ground-truth scores / counts
    / ranks are pre-filled from one un-sharded pass, so the emulated rank ends with the true ranks and metrics (checked).  Timed like
    the headline: two captures of the whole step replayed alternately."""
    from laff_amd import synth
    from laff_amd.dist import EmulatedComm, evaluate_sharded, evaluate_sharded_by_text, shard_bounds
    full = evaluate_sharded(backend, vis, txt, gt, Nt, Nv, heads)
    torch.cuda.synchronize()
    want = tuple(float(x) for x in full['metrics'])
    t0, t1 = shard_bounds(Nt, G, 0)
    v0, v1 = shard_bounds(Nv, G, 0)
    vis_l = {k: synth.slice_rows(v, v0, v1) for k, v in vis.items()}
    txt_l = {k: synth.slice_rows(v, t0, t1) for k, v in txt.items()}
    tmax = max(b - a for a, b in (shard_bounds(Nt, G, r) for r in range(G)))
    vmax = max(b - a for a, b in (shard_bounds(Nv, G, r) for r in range(G)))

    def padded(E, n, nmax):
        out = torch.zeros((G * nmax, E.shape[1]), dtype=E.dtype, device=E.device)
        for r in range(G):
            a, b = shard_bounds(n, G, r)
            out[r * nmax:r * nmax + (b - a)] = E[a:b]
        return out

    state, comm = {}, EmulatedComm(G, 0)
    if scheme == 'text':
        state['gathered_v'] = padded(full['vis_emb'].reshape(Nv, -1), Nv, vmax)
        state['gathered_r'] = padded(full['ranks'].to(torch.int32)[:, None], Nt, tmax).reshape(-1).contiguous()
        fn = evaluate_sharded_by_text
    else:
        state['gathered'] = padded(full['txt_emb'].reshape(Nt, -1), Nt, tmax)
        comm.peers['max'] = full['rank_state'].s_gt64.clone()
        fn = evaluate_sharded
        first = fn(backend, vis_l, txt_l, gt, Nt, Nv, heads, state=state, comm_impl=comm, want_metrics=False)
        comm.peers['sum'] = (full['ranks'] - first['ranks']).to(torch.int32)
        del first
    ranks_full = full['ranks'].clone()
    del full

    def one(slot, **kw):
        return fn(backend, vis_l, txt_l, gt, Nt, Nv, heads, state=state, comm_impl=comm, metrics_out=pins[slot], **kw)
    for slot in (0, 1):
        r = one(slot)
        torch.cuda.synchronize()
        if not torch.equal(r['ranks'], ranks_full) or not np.allclose(pins[slot][:7].numpy(), want, rtol=1e-12, atol=0):
            raise RuntimeError('emulated %s shard 1/%d: ranks / metrics differ from the un-sharded pass' % (scheme, G))
    graphs = [capture_graph(lambda gi=gi: one(gi)) for gi in range(2)]
    for k in range(max(warmup, 8)):
        graphs[k % 2].replay()
    dt = timed_replays(graphs, steps)
    for slot in (0, 1):
        if not np.allclose(pins[slot][:7].numpy(), want, rtol=1e-12, atol=0) or float(pins[slot][7]) != 0.0:
            raise RuntimeError('emulated %s shard 1/%d: a replay left other metrics than the un-sharded pass' % (scheme, G))
    br = instrumented_replays(lambda: one(0), prof, profile_steps)
    rows, cols = (t1 - t0, Nv) if scheme == 'text' else (Nt, v1 - v0)
    out = {'scheme': scheme, 'ranks_of': G, 'rank': 0, 'texts_embedded': t1 - t0, 'videos_embedded': v1 - v0,
           'score_block': [rows, cols], 'ms_per_step': round(1e3 * dt / steps, 4), 'pairs_per_s_this_rank': float(rows) * cols * steps / dt,
           'ideal_ms_at_linear_scaling': None, 'metrics_equal_to_unsharded': True, 'collectives': 'none (peers pre-filled)'}
    if br is not None:
        out['kernels_ms'] = {k: round(v[0], 4) for k, v in br['launches'].items()}
        out['gaps_ms'] = round(br['gaps_ms'], 4)
        out['launches'] = br['n_launches']
    del graphs
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='c4_40kx10k')
    ap.add_argument('--precision', default='fp16', help='similarity GEMM operands: fp16 | fp16x3 | bf16x3 | bf16')
    ap.add_argument('--fc-precision', default='fp16x3', help="FC projections: fp32 (fp32 MFMA) | fp16x3 (exact fp16 hi/lo split)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='time eager launches instead of HIP-graph replays')
    ap.add_argument('--shard', default='video', choices=['auto', 'video', 'text', 'video16'],
                    help="N > 1 decomposition of the headline number: 'video' (default; BASELINE.json's wording: video-row shards, all-gather "
                         "of the text embeddings, two small all-reduces), 'auto' = 'text' / 'video', whichever gathers fewer rows "
                         "(laff_amd.dist.choose_sharding: at 40k x 10k the videos are the smaller side, 2 collective rounds instead of 3), "
                         "'text' (text-row shards, all-gather of the video embeddings, no all-reduce), 'video16' (video-row shards with the "
                         "16-bit text operand + the fp32 video rows gathered and the in-band pairs sent to the owner of their text row).  "
                         "The other schemes are timed too and reported beside it (alt_shards); config.shard names the one `value` is for")
    ap.add_argument('--emulate-shard', type=int, default=0, metavar='G',
                    help="one GPU: time rank 0's share of a G-rank pass of this workload with no collectives (laff_amd.dist.EmulatedComm: "
                         "the peers' gathered rows / reduced values are pre-filled) instead of the whole problem; the scheme is --shard")
    ap.add_argument('--two-streams', action='store_true', help='single GPU: alternate the two captured steps between two streams')
    ap.add_argument('--no-extra-modes', action='store_true', help='skip the sustained loop and the count-only mode (profiling runs)')
    ap.add_argument('--sustain-seconds', type=float, default=2.0, help='extra untimed-by-the-driver loop reporting the sustained rate')
    ap.add_argument('--force-dist', action='store_true', help='run the N > 1 code path (collectives included) on a 1-rank group')
    ap.add_argument('--profile-steps', type=int, default=20, help='replays of the instrumented graph (events between the launches) after the timed region')
    ap.add_argument('--seed', type=int, default=1237)
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        # no launcher: become the parent of N ranks (nothing in this process has touched the GPU yet)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if os.environ.get('LAFF_BENCH_DRYRUN') == '1':
        return dryrun_main(args)

    # stdout carries ONE line (the JSON): native libraries print banners to fd 1 (RCCL's version block at communicator teardown), so
    # fd 1 is pointed at stderr for the whole run and the line is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 or ('RANK' in os.environ and args.force_dist):
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    elif args.force_dist:
        # no launcher environment: a 1-rank RCCL group of our own, so that --force-dist really runs the collectives
        import socket
        sk = socket.socket()
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
        sk.close()
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, world_size=1, rank=0,
                                device_id=torch.device('cuda', local_rank))
    if args.gpus != world:
        if rank == 0:
            print('warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)
    dev = torch.device('cuda', local_rank)

    from laff_amd import synth
    from laff_amd.dist import (HipBackend, check_metrics_flag, choose_sharding, default_pair_bucket_cap, evaluate_sharded,
                               evaluate_sharded_by_text, evaluate_sharded_v16, gathered_bytes, shard_bounds)
    import laff_amd.model.model as M
    M.FC_PRECISION = args.fc_precision
    Nt, Nv, heads, d, frames = synth.WORKLOADS[args.workload]
    spec = synth.SPECS.get(args.workload)
    shard = choose_sharding(Nt, Nv) if args.shard == 'auto' else args.shard
    if shard == 'video16' and args.precision not in ('fp16', 'bf16'):
        raise SystemExit("--shard video16 gathers a one-plane 16-bit operand: --precision must be fp16 or bf16 (got %s)" % args.precision)
    model = synth.build_model(heads, d, dev, frames=frames, seed=args.seed, spec=spec)
    vis, txt, gt, lens = synth.make_features(Nt, Nv, dev, frames=frames, seed=args.seed, spec=spec)
    t0, t1 = shard_bounds(Nt, world, rank)
    v0, v1 = shard_bounds(Nv, world, rank)
    vis_l = {k: synth.slice_rows(v, v0, v1) for k, v in vis.items()}
    txt_l = {k: synth.slice_rows(v, t0, t1) for k, v in txt.items()}
    sparse_nnz = sum(int(v.values().numel()) for v in txt_l.values() if v.layout != torch.strided)
    del vis, txt
    backend = HipBackend(model, args.precision)
    timer = StageTimer()
    from laff_amd import ops
    prof = LaunchProfiler()
    ops.profiler = prof

    if args.emulate_shard > 1:
        if world != 1 or args.force_dist:
            raise SystemExit('--emulate-shard runs on one GPU without a process group')
        pins_e = [torch.zeros(8, dtype=torch.float64).pin_memory() for _ in range(2)]
        scheme = 'text' if shard == 'text' else 'video'
        em = emulate_shard(backend, vis_l, txt_l, gt, Nt, Nv, heads, args.emulate_shard, scheme, args.steps, args.warmup, pins_e, prof,
                           args.profile_steps)
        rows, cols = em['score_block']
        line = {'metric': 'text-video cosine pairs/sec', 'value': em['pairs_per_s_this_rank'], 'unit': 'pairs/s', 'n_gpus': 1, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': em['ms_per_step'], 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
                'dtype': 'f32 towers (FC on fp16 hi/lo split x3 MFMA) + %s similarity' % args.precision, 'data': 'synthetic',
                'config': {'workload': "%s, EMULATED: rank 0's share of a %d-rank '%s'-sharded pass on one GPU, no collectives "
                                       '(towers on %d texts + %d videos, score block %d x %d); value = pairs of THIS block per second'
                                       % (args.workload, args.emulate_shard, scheme, em['texts_embedded'], em['videos_embedded'], rows, cols),
                           'launch': 'HIP graph replay of the whole per-rank step'},
                'quality': {'R@1': float(pins_e[0][0]), 'R@5': float(pins_e[0][1]), 'R@10': float(pins_e[0][2]), 'MedR': float(pins_e[0][3])},
                'shard_emulation': [em], 'roofline': None, 'cpu_baseline': None}
        os.write(json_fd, (json.dumps(line) + '\n').encode())
        return

    metrics_pinned = torch.zeros(8, dtype=torch.float64).pin_memory()
    force_dist = args.force_dist
    if force_dist and not dist.is_initialized():
        raise SystemExit('--force-dist: no process group could be initialised')
    distributed = world > 1 or force_dist

    pins = [metrics_pinned, torch.zeros(8, dtype=torch.float64).pin_memory()]

    def step(timed, async_metrics=False, runner=None, state=None, slot=0, want_scores=True, kind=None):
        try:
            return step_(timed, async_metrics, runner, state, slot, want_scores, kind)
        finally:
            timer.enabled = prof.enabled = False      # (a capture that follows must not record plain events)

    def step_(timed, async_metrics, runner, state, slot, want_scores, kind):
        timer.enabled = timed
        prof.enabled = timed
        timer.start()
        if (kind or shard) == 'text' and distributed:
            return evaluate_sharded_by_text(backend, vis_l, txt_l, gt, Nt, Nv, heads, timer=timer, want_scores=want_scores,
                                            metrics_out=pins[slot] if async_metrics else None, runner=runner, state=state,
                                            force_collectives=force_dist, finish_tag=str(slot))
        if (kind or shard) == 'video16' and distributed:
            return evaluate_sharded_v16(backend, vis_l, txt_l, gt, Nt, Nv, heads, timer=timer, want_scores=want_scores,
                                        metrics_out=pins[slot] if async_metrics else None, runner=runner, state=state,
                                        force_collectives=force_dist, finish_tag=str(slot))
        return evaluate_sharded(backend, vis_l, txt_l, gt, Nt, Nv, heads, timer=timer, want_scores=want_scores,
                                metrics_out=pins[slot] if async_metrics else None, runner=runner, state=state,
                                force_collectives=force_dist, finish_tag=str(slot))

    power_w = []

    def sclk_sampler(stop, out):
        # shader clock actually held during a loop, and the package power beside it (the chip clocks to its power budget:
        # MI355X_MICROARCH.md "DVFS give-back"; at 1,400 W the pass is bound by its energy, not by any one pipe)
        import re
        import subprocess
        while not stop.is_set():
            try:
                txt_ = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=5).stdout
                m = re.findall(r'sclk clock level:? *\d*:? *\((\d+)Mhz\)', txt_)
                if m:
                    out.append(int(m[local_rank if local_rank < len(m) else 0]))
                pw = re.findall(r'Package Power \(W\): *([0-9.]+)', txt_)
                if pw:
                    power_w.append(float(pw[local_rank if local_rank < len(pw) else 0]))
            except Exception:  # noqa: BLE001
                return

    # Launch modes of the timed region (host issue of ~15 launches costs 0.3-0.4 ms per step, as much as the GPU work of a
    # 1/4 shard, and sits on the critical path because every step ends with a host sync on the 7 metrics):
    #   single GPU : ONE captured HIP graph per step;
    #   N > 1      : one captured graph per LOCAL phase (laff_amd.dist.GraphRunner), the three RCCL collectives eager
    #                between them -- nothing RCCL-related is ever captured;
    #   --no-graph : eager launches.
    graph, graphs, runner, state, side = None, [], None, {}, None
    for _ in range(max(1, args.warmup)):
        res = step(False)
    torch.cuda.synchronize()
    eager_metrics = tuple(float(x) for x in res['metrics']) if res.get('metrics') is not None else None
    if not args.no_graph:
        try:
            if distributed:
                from laff_amd.dist import GraphRunner
                runner = GraphRunner()
                for slot in (0, 1):      # captures + runs every phase once (the 'finish' phase once per metrics buffer)
                    res = step(False, async_metrics=True, runner=runner, state=state, slot=slot)
                    torch.cuda.synchronize()
            else:
                # two captures of the same step with their own output buffers: step k+1 is enqueued while the host still
                # reads the 7 metrics of step k, so the GPU never idles between steps (a single graph with a host sync after
                # every replay left a 30 us hole per step = 2.4 %)
                graphs = []
                for gi in range(2):
                    gph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gph, capture_error_mode='thread_local'):
                        res = evaluate_sharded(backend, vis_l, txt_l, gt, Nt, Nv, heads, metrics_out=pins[gi])
                    gph.replay()
                    torch.cuda.synchronize()
                    graphs.append(gph)
                graph = graphs[0]
                if args.two_streams:
                    # steps k and k+1 on two streams: the latency-bound tail of a step (prepare / resolve / metrics, the sparse last
                    # round of a GEMM) runs beside the next step's first kernels; the two captures share read-only inputs only (each
                    # captured laff_rank_metrics_async call has its own device result slot)
                    side = [torch.cuda.Stream(), torch.cuda.Stream()]
        except Exception as e:  # noqa: BLE001
            print('warning: HIP graph capture failed (%s); timing eager launches' % e, file=sys.stderr)
            graph, runner = None, None
            torch.cuda.synchronize()
            res = step(False)
    launch_mode = ('HIP graph replay, step k+1 enqueued while the host reads the metrics of step k; the two captures are checked against the eager step (%d pipelined replays, every one compared) before the W warm-up replays' % GRAPH_CHECK_REPLAYS if graph is not None else
                   ('per-phase HIP graphs + eager RCCL, step k+1 issued while the host reads the metrics of step k'
                    if runner is not None else 'eager'))

    def timed_step():
        if graph is not None:
            graph.replay()
            torch.cuda.current_stream().synchronize()     # the step's result (7 metrics) is on the host
            return None
        if runner is not None:
            r = step(False, async_metrics=True, runner=runner, state=state)
            torch.cuda.current_stream().synchronize()
            return r
        return step(True)        # eager: per-launch events are recorded inside the timed region

    def dist_loop(kind, runner_, state_):
        """K steps of one N > 1 decomposition, two in flight at most; returns the wall time (barrier + sync on both sides, MAX over ranks)."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        ev = [torch.cuda.Event(), torch.cuda.Event()]
        for k in range(args.steps):
            if runner_ is not None:
                step(False, async_metrics=True, runner=runner_, state=state_, slot=k % 2, kind=kind)
                ev[k % 2].record()
                if k:
                    ev[(k - 1) % 2].synchronize()
                    check_metrics_flag(pins[(k - 1) % 2])
            else:
                step(False, kind=kind)
        torch.cuda.synchronize()
        if runner_ is not None:
            check_metrics_flag(pins[(args.steps - 1) % 2])
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - ts
        if world > 1:
            tt_ = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            el = float(tt_.item())
        return el

    # The W warm-up steps proper: the SAME steps as the timed ones (graph replays when the step is a graph), right in front of the
    # timed region.  (The eager steps above only fill the allocator before the capture; the capture itself is host work during which
    # the GPU idles and drops its clocks -- with the warm-up in front of it the first timed steps ran on a chip still ramping up:
    # 1.19 ms per step over 20 steps against 1.115 sustained.)
    if graph is not None:
        # graph self-check: both captures, replayed alternately the way the timed loop replays them (two in flight, the host reads the
        # metrics of replay k - 1 while replay k runs), must reproduce the eager step's seven metrics EVERY time.  GRAPH_CHECK_REPLAYS
        # replays = ~60 ms of device work; they also bring the chip out of the state the capture (host work, idle GPU) leaves it in -- the
        # package needs tens of milliseconds of load to settle at its power cap, and the first steps behind an idle phase run 3-5 %
        # slow: LAFF_BENCH_STEP_TRACE=1 prints the per-step intervals.
        want = eager_metrics
        ev_c = [torch.cuda.Event(), torch.cuda.Event()]

        def check_replay(k):
            ev_c[k % 2].synchronize()
            check_metrics_flag(pins[k % 2])
            got = tuple(pins[k % 2][:7].tolist())
            if want is not None and got != want:
                raise RuntimeError('graph replay %d gave metrics %s, the eager step %s' % (k, got, want))
        for k in range(GRAPH_CHECK_REPLAYS):
            graphs[k % 2].replay()
            ev_c[k % 2].record()
            if k:
                check_replay(k - 1)
        check_replay(GRAPH_CHECK_REPLAYS - 1)
    for k in range(args.warmup):
        if graph is not None:
            graphs[k % 2].replay()
        elif runner is not None:
            step(False, async_metrics=True, runner=runner, state=state, slot=k % 2)
        else:
            step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    if graph is not None or runner is not None:
        # two steps in flight at most: step k is issued, then the host waits for step k-1 and reads ITS metrics buffer
        events = [torch.cuda.Event(), torch.cuda.Event()]
        seen = []
        step_trace = [] if os.environ.get('LAFF_BENCH_STEP_TRACE') else None      # debugging: host time at which every step's result arrived
        for k in range(args.steps):
            if graph is not None and side is not None:
                with torch.cuda.stream(side[k % 2]):
                    graphs[k % 2].replay()
                    events[k % 2].record()
            elif graph is not None:
                graphs[k % 2].replay()
                events[k % 2].record()
            else:
                res = step(False, async_metrics=True, runner=runner, state=state, slot=k % 2)
                events[k % 2].record()
            if k:
                events[(k - 1) % 2].synchronize()
                if step_trace is not None:
                    step_trace.append(time.perf_counter())
                check_metrics_flag(pins[(k - 1) % 2])
                seen.append(float(pins[(k - 1) % 2][0]))
        events[(args.steps - 1) % 2].synchronize()
        check_metrics_flag(pins[(args.steps - 1) % 2])
        seen.append(float(pins[(args.steps - 1) % 2][0]))
        assert len(seen) == args.steps
    else:
        for _ in range(args.steps):
            r = timed_step()
            if r is not None:
                res = r
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if (graph is not None or runner is not None) and step_trace:
        print('step completion intervals (us): ' + ' '.join('%.0f' % (1e6 * (b - a)) for a, b in zip([t_start] + step_trace[:-1], step_trace)), file=sys.stderr)
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    breakdown = None
    if graph is not None or runner is not None:
        final_metrics = tuple(pins[(args.steps - 1) % 2][:7].tolist())
        if graph is not None:
            # per-kernel durations of WHAT IS TIMED: the same step captured a third time with an event-record node in front of and behind
            # every launch, replayed --profile-steps times (the events cost a few microseconds per step: `instrumented_span_ms` against
            # `ms_per_step`)
            breakdown = instrumented_replays(
                lambda: evaluate_sharded(backend, vis_l, txt_l, gt, Nt, Nv, heads, metrics_out=pins[0]), prof, args.profile_steps)
        prof_steps = args.profile_steps
        if breakdown is None:
            # fallback (N > 1: the collectives stay eager between per-phase graphs; or a stack without event nodes): the same kernels
            # on the same data launched eagerly with events around each launch.  One eager step first, without events: it
            # re-allocates what the graphs' private pools held
            res = step(False)
            torch.cuda.synchronize()
            prof_steps = min(args.profile_steps, 5)
            for _ in range(prof_steps):
                res = step(True)
            torch.cuda.synchronize()
    else:
        final_metrics = res['metrics']
        prof_steps = args.steps

    # the other N > 1 decompositions, timed the same way, reported beside the headline one
    alt, alts = None, []
    if distributed:
        K_emb = heads * d
        for other in [k for k in ('video', 'text', 'video16') if k != shard and (k != 'video16' or backend.v16_ok())]:
            try:
                r2, s2 = None, {}
                if runner is not None:
                    from laff_amd.dist import GraphRunner
                    r2 = GraphRunner()
                    for slot in (0, 1):
                        step(False, async_metrics=True, runner=r2, state=s2, slot=slot, kind=other)
                        torch.cuda.synchronize()
                else:
                    step(False, kind=other)
                el2 = dist_loop(other, r2, s2)
                one = {'shard': other, 'ms_per_step': 1e3 * el2 / args.steps, 'value': float(Nt) * Nv * args.steps / el2,
                       'R@1': float(pins[(args.steps - 1) % 2][0]) if r2 is not None else None,
                       'gathered_bytes_per_step': gathered_bytes(other, Nt, Nv, K_emb, world, default_pair_bucket_cap(Nt, world))}
                del r2, s2
            except Exception as e:  # noqa: BLE001  (every rank takes the same path: the collectives stay matched)
                one = {'shard': other, 'error': str(e)}
            alts.append(one)
        alt = alts[0] if alts else None

    sustained, no_scores, strict, shard_emul = None, None, None, None
    if graph is not None and world == 1 and not distributed and not args.no_extra_modes:
        # (i) sustained rate: the driver's timed region is tens of milliseconds -- a burst; this loop runs >= --sustain-seconds
        if args.sustain_seconds > 0:
            import threading
            stop, clocks = threading.Event(), []
            th = threading.Thread(target=sclk_sampler, args=(stop, clocks), daemon=True)
            th.start()
            torch.cuda.synchronize()
            ts, n_s = time.perf_counter(), 0
            while time.perf_counter() - ts < args.sustain_seconds:
                for k in range(50):
                    graphs[k % 2].replay()
                torch.cuda.current_stream().synchronize()
                n_s += 50
            dt_s = time.perf_counter() - ts
            stop.set()
            th.join(timeout=6)
            sustained = {'seconds': round(dt_s, 2), 'steps': n_s, 'ms_per_step': round(1e3 * dt_s / n_s, 4),
                         'value': float(Nt) * Nv * n_s / dt_s, 'sclk_mhz_samples': clocks[:8],
                         # (the first sample is taken while the loop starts; the rest sit at the package cap: the pass is energy-bound)
                         'package_power_w_samples': power_w[:8],
                         'joules_per_step': (round(1e-3 * dt_s / n_s * (sum(power_w[1:8]) / len(power_w[1:8])) * 1e3, 4)
                                             if len(power_w) > 1 else None)}
        # (ii) the count-only mode (ranks + metrics, S never written): what the similarity GEMM does when the caller does not ask for
        # the score matrix (with K = 512 the fp32 S store is the HBM-bound part; without it the GEMM is MFMA-bound)
        g2 = []
        prof.enabled = timer.enabled = False          # no events inside a capture
        for gi in range(2):
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, capture_error_mode='thread_local'):
                evaluate_sharded(backend, vis_l, txt_l, gt, Nt, Nv, heads, metrics_out=pins[gi], want_scores=False)
            gph.replay()
            torch.cuda.synchronize()
            g2.append(gph)
        tn = time.perf_counter()
        for k in range(args.steps):
            g2[k % 2].replay()
        torch.cuda.synchronize()
        dt_n = time.perf_counter() - tn
        check_metrics_flag(pins[0]); check_metrics_flag(pins[1])
        m_ns = tuple(pins[(args.steps - 1) % 2][:7].tolist())
        br_ns = instrumented_replays(lambda: evaluate_sharded(backend, vis_l, txt_l, gt, Nt, Nv, heads, metrics_out=pins[0],
                                                               want_scores=False), prof, args.profile_steps)
        if br_ns is not None:
            sim_ns_ms = br_ns['launches'].get('sim_gemm', (0.0, 1))[0]
        else:
            prof.spans, keep_spans = [], prof.spans
            for _ in range(5):
                step(True, want_scores=False)
            torch.cuda.synchronize()
            sim_ns = prof.totals().get('sim_gemm', (0.0, 1))
            sim_ns_ms = sim_ns[0] / max(1, sim_ns[1])
            prof.spans = keep_spans
        Kk = heads * d
        fl = 2.0 * Kk * float(Nt) * Nv * (3 if args.precision.endswith('x3') else 1)
        no_scores = {'ms_per_step': round(1e3 * dt_n / args.steps, 4), 'value': float(Nt) * Nv * args.steps / dt_n,
                     'metrics_equal_to_headline_mode': m_ns == tuple(final_metrics),
                     # every launch of the count-only step (stamps inside a capture of it, like `kernels` for the headline mode): the
                     # GEMM is ~0.09 ms shorter than with S but the step only ~0.03: with no 1.6 GB of fresh score lines between them,
                     # the launches behind the GEMM (rank_resolve) and the next step's first ones run on a chip the GEMM left hotter --
                     # the pass is energy-bound (profiles/r6_energy*.json: 1.37 J vs 1.20 J per step, both at the package cap)
                     'kernels_ms': ({k: round(v[0], 4) for k, v in br_ns['launches'].items()} if br_ns is not None else None),
                     'gaps_ms': (round(br_ns['gaps_ms'], 4) if br_ns is not None else None),
                     'roofline': {'kernel': 'laff_sim_gemm_banded (S == NULL)', 'bound': 'mfma', 'avg_launch_ms': round(sim_ns_ms, 5),
                                  'achieved': round(fl / (sim_ns_ms * 1e-3) / 1e12, 2) if sim_ns_ms > 0 else None,
                                  'peak': MFMA_PEAK_TFLOPS['f16'], 'unit': 'TFLOP/s',
                                  'frac': round(fl / (sim_ns_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS['f16'], 4) if sim_ns_ms > 0 else None}}
        del g2
        # (iii) the strict mode: the same pass with the hi/lo split similarity (fp16x3: fp32-class scores, what model.predict() defaults to)
        try:
            b3 = HipBackend(model, 'fp16x3')
            evaluate_sharded(b3, vis_l, txt_l, gt, Nt, Nv, heads, metrics_out=pins[0])
            torch.cuda.synchronize()
            g3 = [capture_graph(lambda gi=gi: evaluate_sharded(b3, vis_l, txt_l, gt, Nt, Nv, heads, metrics_out=pins[gi])) for gi in range(2)]
            for k in range(8):
                g3[k % 2].replay()
            dt_3 = timed_replays(g3, args.steps)
            check_metrics_flag(pins[0]); check_metrics_flag(pins[1])
            strict = {'precision': 'fp16x3', 'ms_per_step': round(1e3 * dt_3 / args.steps, 4), 'value': float(Nt) * Nv * args.steps / dt_3,
                      'metrics_equal_to_headline_mode': tuple(pins[(args.steps - 1) % 2][:7].tolist()) == tuple(final_metrics)}
            del g3, b3
        except Exception as e:  # noqa: BLE001
            strict = {'precision': 'fp16x3', 'error': str(e)}
        # (iv) what one rank of an 8-GPU run of this workload would execute (no collectives): small-shape efficiency of the same kernels
        shard_emul = []
        for scheme in ('text', 'video'):
            try:
                shard_emul.append(emulate_shard(backend, vis_l, txt_l, gt, Nt, Nv, heads, 8, scheme, args.steps, args.warmup, pins, prof,
                                                args.profile_steps))
            except Exception as e:  # noqa: BLE001
                shard_emul.append({'scheme': scheme, 'ranks_of': 8, 'error': str(e)})
        for e_ in shard_emul:
            if 'ms_per_step' in e_:
                e_['ideal_ms_at_linear_scaling'] = round(1e3 * elapsed / args.steps / 8, 4)

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        pairs = float(Nt) * Nv
        stages = {k: t / c for k, (t, c) in timer.totals().items()}
        res = dict(res, metrics=final_metrics)     # ms per step per stage (this rank)
        # ---- roofline of the dominant kernel: ALGORITHMIC work (SURVEY.md section 8d / DESIGN.md) / measured launch time
        K = heads * d
        nvl, ntl = v1 - v0, t1 - t0
        if spec is None:
            fc_v, fc_t, raw_v, raw_t, gather_dims = [512] * 4, [512] * 4, 0, 0, []
        else:    # FC'd dense features, no-transform (raw) feature widths, sparse (gather) vocabularies
            fc_v = [dk for n, dk in spec['vid'].items() if n not in spec['vis_no_transform']]
            raw_v = sum(dk for n, dk in spec['vid'].items() if n in spec['vis_no_transform'])
            fc_t, raw_t, gather_dims = [], spec['txt']['CLIP'], [spec['txt']['bow']]
        fc_macs = float(K) * (nvl * sum(fc_v) + ntl * sum(fc_t))
        Lv, Lt = len(fc_v) + (1 if raw_v else 0), len(fc_t) + len(gather_dims) + (1 if raw_t else 0)
        if breakdown is not None:
            launches = dict(breakdown['launches'])                                                    # ms per step, launches per step
        else:
            launches = {k: (t / prof_steps, c // prof_steps) for k, (t, c) in prof.totals().items()}
        listed_n = None
        try:
            if res.get('rank_state') is not None and world == 1:
                listed_n = res['rank_state'].listed_pairs()[0]
        except Exception:  # noqa: BLE001
            listed_n = None
        x3 = 3 if args.precision.endswith('x3') else 1
        work = {   # entry point -> (bound, algorithmic units per step on this rank, peak, unit scale)
            'fc_act_bn': ('mfma', 2.0 * fc_macs * (3 if args.fc_precision == 'fp16x3' else 1),
                          MFMA_PEAK_TFLOPS['f16' if args.fc_precision == 'fp16x3' else 'f32'], 1e12, 'TFLOP/s'),
            'split_rows': ('hbm', (4.0 + 4.0) * (nvl * sum(fc_v) + ntl * sum(fc_t)), HBM_PEAK_GBS, 1e9, 'GB/s'),
            'row_scales': ('hbm', 4.0 * (nvl * sum(fc_v) + ntl * sum(fc_t)), HBM_PEAK_GBS, 1e9, 'GB/s'),
            'fuse': ('hbm', 4.0 * K * (nvl * (Lv + 1 - (1 if raw_v else 0)) + ntl * (Lt + 1 - (1 if raw_t else 0)))
                     + 4.0 * (nvl * raw_v + ntl * raw_t), HBM_PEAK_GBS, 1e9, 'GB/s'),
            'frame_fuse': ('hbm', 4.0 * d * 4 * ((float(lens[v0:v1].sum()) if lens is not None else 0.0) + nvl), HBM_PEAK_GBS, 1e9, 'GB/s'),
            'fc_gather': ('hbm', 4.0 * K * (sum(gather_dims) + ntl) + 8.0 * sparse_nnz, HBM_PEAK_GBS, 1e9, 'GB/s'),
            'pack_rows': ('hbm', (4.0 + 2.0 * (2 if x3 == 3 else 1)) * (ntl + nvl) * K, HBM_PEAK_GBS, 1e9, 'GB/s'),
            'rank_count': ('hbm', 4.0 * Nt * nvl, HBM_PEAK_GBS, 1e9, 'GB/s'),
            'gather_gt': ('hbm', 8.0 * Nt, HBM_PEAK_GBS, 1e9, 'GB/s'),
            'row_dot_gt': ('hbm', 2.0 * K * (Nt + min(Nt, nvl)), HBM_PEAK_GBS, 1e9, 'GB/s'),
            # embeddings (fp32) + operands (16-bit) of every row once, + the ground-truth video row of every text (at most nvl distinct)
            'rank_prepare': ('hbm', (4.0 + 2.0 * (2 if x3 == 3 else 1)) * K * (Nt + nvl) + 4.0 * K * min(Nt, nvl), HBM_PEAK_GBS, 1e9, 'GB/s'),
            # the two fp32 rows of every listed pair (8 K bytes per pair; they come out of L2 / Infinity Cache, the
            # HBM peak is only the common denominator) -- latency-bound launches: the fraction says how far from a streaming kernel
            # priced against the L2's aggregate rate (MI355X_MICROARCH.md: ~34.5 TB/s), not HBM: the rows of a listed pair are re-read
            # out of L2 / Infinity Cache (C3: 304k pairs x 4 KB = 1.2 GB for 26 MB of distinct rows) -- against the HBM peak the same
            # figure exceeded 1 (round 5's C3 line)
            'rank_resolve': ('l2', 8.0 * K * (listed_n or 0) + 4.0 * Nt, L2_PEAK_GBS, 1e9, 'GB/s'),
            'rank_export': ('l2', 8.0 * (listed_n or 0) + 4.0 * Nt, L2_PEAK_GBS, 1e9, 'GB/s'),
            # a chain of dependent device-scope round trips over <= 256 workgroups: no bandwidth or flop roof applies
            'rank_metrics': ('latency', 0.0, None, 1.0, None),
            'plane_row_norms': ('hbm', 4.0 * K * (nvl * Lv + ntl * Lt), HBM_PEAK_GBS, 1e9, 'GB/s'),
        }
        sim_r, sim_c = (ntl, Nv) if (shard == 'text' and distributed) else (Nt, nvl)
        sim_bytes = 4.0 * sim_r * sim_c + 2.0 * (sim_r + sim_c) * K              # fp32 S written once + 16-bit operands read once
        sim_flops = 2.0 * K * sim_r * sim_c * x3
        sim_ms = launches.get('sim_gemm', (0.0, 0))[0]
        if sim_ms > 0:
            hbm_frac = sim_bytes / (sim_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            mfma_frac = sim_flops / (sim_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS['f16']
            if mfma_frac >= hbm_frac:
                work['sim_gemm'] = ('mfma', sim_flops, MFMA_PEAK_TFLOPS['f16'], 1e12, 'TFLOP/s')
            else:
                work['sim_gemm'] = ('hbm', sim_bytes, HBM_PEAK_GBS, 1e9, 'GB/s')
        dom = max(launches, key=lambda k: launches[k][0]) if launches else None
        roof = {'bound': None, 'achieved': None, 'peak': None, 'unit': None, 'frac': None, 'traffic': None}
        per_kernel = {}
        for k, (ms, n) in launches.items():
            if k in work and ms > 0:
                bound, units, peak, scale, unit = work[k]
                if peak is None:
                    per_kernel[k] = {'ms_per_step': round(ms, 4), 'launches_per_step': n, 'bound': bound, 'achieved': None, 'peak': None,
                                     'unit': None, 'frac': None}
                    continue
                ach = units / (ms * 1e-3) / scale
                per_kernel[k] = {'ms_per_step': round(ms, 4), 'launches_per_step': n, 'bound': bound, 'achieved': round(ach, 2),
                                 'peak': peak, 'unit': unit, 'frac': round(ach / peak, 4)}
        if 'fc_act_bn' in per_kernel and args.fc_precision == 'fp16x3':
            # `achieved` counts the flops the fp16 pipe EXECUTES (three hi/lo products per MAC); SURVEY section 8d's algorithmic figure
            # is 2 N D_k D: both fractions, so that neither reading is hidden
            pk = per_kernel['fc_act_bn']
            alg = 2.0 * fc_macs / (pk['ms_per_step'] * 1e-3) / 1e12
            pk['flops_counted'] = 'executed: 3 fp16 MFMA products per algorithmic MAC (fp32-class accuracy on the fp16 pipe)'
            pk['achieved_algorithmic'] = round(alg, 2)
            pk['frac_algorithmic'] = round(alg / MFMA_PEAK_TFLOPS['f16'], 4)
            # (not a fraction of a roof: how many times the fp32 matrix pipe's PEAK the algorithmic rate is -- why the FC runs as three
            # fp16 products instead of on v_mfma_f32_32x32x2_f32)
            pk['algorithmic_rate_over_f32_mfma_peak'] = round(alg / MFMA_PEAK_TFLOPS['f32'], 3)
        if 'fuse' in per_kernel:
            # `achieved` = SURVEY section 8d's bytes (planes in, embeddings out).  The launches also write the 16-bit GEMM operand and,
            # with laff_fuse_packed_rank, read every text's ground-truth video row (L2 / Infinity Cache): what they actually move
            esz = 2.0 * (2 if x3 == 3 else 1)
            moved = work['fuse'][1] + esz * K * (nvl + ntl) + (4.0 * K * ntl if world == 1 else 0.0)
            per_kernel['fuse']['achieved_incl_operand_and_gt_rows'] = round(moved / (per_kernel['fuse']['ms_per_step'] * 1e-3) / 1e9, 2)
        if dom in per_kernel:
            pk = per_kernel[dom]
            roof = {'kernel': 'laff_' + dom, 'bound': pk['bound'], 'achieved': pk['achieved'], 'peak': pk['peak'], 'unit': pk['unit'],
                    'frac': pk['frac'], 'traffic': None, 'launches_per_step': pk['launches_per_step'],
                    'avg_launch_ms': round(pk['ms_per_step'] / max(1, pk['launches_per_step']), 5)}
        # measured HBM traffic of the dominant kernel: rocprofv3 PMC passes of this same command (tools/profile_round.sh), committed
        # under profiles/ and keyed to the kernel sources they were measured on -- a stale file is refused, not quoted
        try:
            import glob
            from laff_amd.build import source_hash
            sha = source_hash()
            for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')), reverse=True):
                tj = json.load(open(f))
                if (tj.get('src_sha') == sha and tj.get('workload') == args.workload and tj.get('precision') == args.precision and
                        tj.get('fc_precision') == args.fc_precision and dom in tj['kernels'] and world == 1):
                    k = tj['kernels'][dom]
                    roof['traffic'] = round((k['fetch_corrected_MB'] + k['write_MB']) * 1e6)
                    alg = sim_bytes if dom == 'sim_gemm' else (work[dom][1] if work.get(dom, ('', 0))[0] == 'hbm' else None)
                    roof['traffic_note'] = ('bytes per launch: WRITE_SIZE + 2 x FETCH_SIZE (gfx950 correction), rocprofv3 --pmc passes of '
                                            'this command on these kernel sources (src_sha %s), %s' % (sha, os.path.basename(f))
                                            + ('; algorithmic %.0f MB' % (alg / 1e6) if alg else ''))
                    break
            else:
                roof['traffic_note'] = ('no PMC measurement under profiles/ matches this run: needs kernel sources src_sha %s, workload %s, '
                                        'precision %s / %s, one GPU' % (sha, args.workload, args.precision, args.fc_precision))
        except Exception as e:  # noqa: BLE001
            roof['traffic_note'] = 'traffic lookup failed: %s' % e
        # the denominators are the NOMINAL peaks of MI355X_MICROARCH.md; what the chip sustains under this step is lower
        roof['peak_note'] = ('nominal peaks: 8 TB/s HBM3E (6.29 TB/s measured copy rate), 2.5 PFLOP/s dense fp16 at 2.4 GHz.  The step runs '
                             'at the 1,400 W package cap: held shader clock %s MHz (sustained.sclk_mhz_samples), i.e. an MFMA ceiling of '
                             '~%.1f PFLOP/s at the clock actually held'
                             % (('%d-%d' % (min(sustained['sclk_mhz_samples']), max(sustained['sclk_mhz_samples'])))
                                if sustained and sustained.get('sclk_mhz_samples') else 'unsampled',
                                2.5 * (sum(sustained['sclk_mhz_samples']) / len(sustained['sclk_mhz_samples']) / 2400.0)
                                if sustained and sustained.get('sclk_mhz_samples') else 2.5))
        m = res['metrics']
        agreement = None
        if world == 1 and not args.no_cpu_baseline:
            # outside the timed region: (i) the same embeddings through the strict (hi/lo split, ~1e-7) similarity with its own exact-rank
            # pass, (ii) ranks of float64 scores of the same embeddings computed with torch on the device, in row blocks
            from laff_amd import ops as _ops
            te, ve = res['txt_emb'], res['vis_emb']
            Ts = _ops.pack_rows(te, True, 1e-13, 'fp16x3')
            Vs = _ops.pack_rows(ve, True, 1e-13, 'fp16x3')
            Ss, cs, _st = _ops.exact_ranks(te, ve, Ts, Vs, gt)
            ms = _ops.rank_metrics(cs, base=1)
            t3 = te.reshape(Nt, heads, -1).double()
            v3 = ve.reshape(Nv, heads, -1).double()
            v3 = v3 / (v3.pow(2).sum(2, keepdim=True).sqrt() + (1e-13 + 1e-14))
            r64 = torch.empty(Nt, dtype=torch.int32, device=dev)
            for a in range(0, Nt, 4096):
                tb = t3[a:a + 4096]
                tb = tb / (tb.pow(2).sum(2, keepdim=True).sqrt() + (1e-13 + 1e-14))
                S64 = torch.einsum('thd,vhd->tv', tb, v3) / heads
                g = gt[a:a + 4096].long()
                sg = S64.gather(1, g[:, None])
                ab = S64 > sg
                ab[torch.arange(ab.shape[0], device=dev), g] = False
                r64[a:a + 4096] = ab.sum(1).to(torch.int32) + 1
            listed = res['rank_state'].listed_pairs() if res.get('rank_state') is not None else (None, None)
            agreement = {'max_abs_score_diff_vs_fp16x3': float((Ss - res['S_local']).abs().max()) if res.get('S_local') is not None else None,
                         'identical_ranks_frac': float(((cs + 1) == res['ranks']).float().mean()),
                         'identical_ranks_frac_vs_fp64_scores': float((r64 == res['ranks']).float().mean()),
                         'strict_R@1/5/10/MedR': [ms[0], ms[1], ms[2], ms[3]],
                         'pairs_inside_error_band': listed[0], 'pair_list_overflow': listed[1]}
            del Ss, Ts, Vs, t3, v3
        line = {
            'metric': 'text-video cosine pairs/sec', 'value': pairs / elapsed * args.steps, 'unit': 'pairs/s',
            'n_gpus': world, 'rccl_ranks': dist.get_world_size() if dist.is_initialized() else 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_step,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32 towers (FC on %s) + %s similarity' % ('fp32 MFMA' if args.fc_precision == 'fp32' else 'fp16 hi/lo split x3 MFMA', args.precision),
            'data': 'synthetic',
            'config': {'workload': '%s: %d texts x %d videos, %s, %d head(s) x d=%d' % (
                           args.workload, Nt, Nv, '4+4 features of 512-d' if spec is None else
                           'video %s (no FC: %s) + text %s (no FC: %s, bow sparse CSR)' % (spec['vid'], spec['vis_no_transform'],
                                                                                         spec['txt'], spec['txt_no_transform']),
                           heads, d),
                       'parallelism': ({'video': 'video-row shards x%d, all-gather of the fp32 text embeddings, all-reduce MAX(s_gt) + SUM(counts)',
                                        'text': 'text-row shards x%d, all-gather of the fp32 video embeddings + of the ranks',
                                        'video16': 'video-row shards x%d, all-gather of the 16-bit text operand + of the fp32 video embeddings + of '
                                                   '{s_gt64, band}, all-to-all of the in-band pairs to the text owners, all-reduce SUM(counts), '
                                                   'all-gather of the ranks'}[shard] % world)
                       if distributed else 'single GPU',
                       'force_dist': bool(force_dist), 'shard': shard if distributed else None,
                       'scores': 'fp32 S materialised in HBM (count-only mode reported under no_scores_mode)',
                       'ranks': 'exact: error-band count in the GEMM epilogue + fp64 re-score of the in-band pairs',
                       'launch': launch_mode},
            'quality': {'R@1': m[0], 'R@5': m[1], 'R@10': m[2], 'MedR': m[3], 'meanr': m[4], 'mir': m[5], 'mAP': m[6],
                        'vs_strict_similarity': agreement},
            # per-step time this rank's stream spent in / waiting on each collective (eager pass, HIP events around the call)
            'collective_ms': ({k: round(stages[k], 4) for k in ('all_gather_wait', 'allreduce_s_gt', 'allreduce_count', 'allgather_ranks',
                                                                  'allgather_sgt_band', 'alltoall_pairs') if k in stages} if distributed else None),
            # payload bytes that reach one rank per step from the others, by collective (laff_amd.dist.gathered_bytes)
            'gathered_bytes_per_step': (gathered_bytes(shard, Nt, Nv, heads * d, world, default_pair_bucket_cap(Nt, world))
                                        if distributed else None),
            'alt_shard': alt,
            'alt_shards': alts if distributed else None,
            # (only when the per-kernel figures had to come from an eager pass: host-issued launches, longer than a graph step)
            'stages_ms_eager_pass': ({k: round(v, 4) for k, v in stages.items()} if breakdown is None else None),
            'sustained': sustained,
            'no_scores_mode': no_scores,
            'strict_mode': strict,
            'shard_emulation': shard_emul,
            'kernels': per_kernel,
            # where `kernels` comes from and how it adds up: launches + the idle time between them = the instrumented replay's span
            'kernels_source': ('device wall-clock stamps (one-thread laff_stamp launches) in front of and behind every launch inside a third '
                               'capture of the timed step, mean of %d samples (each the last of a run of back-to-back replays, ~40 ms of continuous load: '
                               'the clocks of the timed region), one stamp interval subtracted per launch (each interval holds '
                               'two launch gaps, so kernels_ms carries ~ one gap per launch that gaps_ms then lacks: the sum is exact).  '
                               'AUTHORITATIVE for A/B and for roofline.achieved: these stamps -- the graph the timed region replays, at '
                               'the clocks it holds; the rocprofv3 kernel-trace averages under profiles/ come from a profiled run of the '
                               'same command (2-3 %% lower clocks) and agree within that' % args.profile_steps
                               if breakdown is not None else 'eager launches with HIP events around each C-ABI call (%d steps)' % prof_steps),
            # kernels_ms + launches x stamp_cost_ms + gaps_ms = instrumented_span_ms (by construction); the un-instrumented step is ms_per_step
            'step_breakdown': ({'kernels_ms': round(sum(v[0] for v in breakdown['launches'].values()), 4),
                                'gaps_ms': round(breakdown['gaps_ms'], 4), 'instrumented_span_ms': round(breakdown['span_ms'], 4),
                                'launches': breakdown['n_launches'], 'stamp_cost_ms': round(breakdown['stamp_cost_ms'], 5),
                                'ms_per_step': round(ms_step, 4)} if breakdown is not None else None),
            'roofline': roof,
        }
        if not args.no_cpu_baseline and world == 1:
            sn, sv = (Nt, Nv) if float(Nt) * Nv <= 4.0e8 else (40000, 10000)       # ~10-20 s of host work
            line['cpu_baseline'] = cpu_baseline(args.workload, sn, sv, heads, d, args.seed, spec)
        os.write(json_fd, (json.dumps(line) + '\n').encode())
    # orderly teardown: captured graphs and their private pools go away while the HIP runtime is still up
    import gc
    graph = graphs = runner = state = res = None
    gc.collect()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
