"""Multi-GPU decomposition of the hot path (one process per GPU, torch.distributed; backend 'nccl' is RCCL over xGMI).

The reference has no parallelism of any kind; this is the one data-parallel decomposition the
path admits:

  * video rows are split contiguously: rank g owns videos [v0, v1) end to end (features -> embeddings -> its
    column block S[:, v0:v1] of the score matrix);
  * text rows are split the same way for the embedding stage only, then ONE all-gather of the text GEMM operand
    (Nt x K 16-bit) gives every rank all texts; it is issued asynchronously and overlaps the video tower;
  * ranks: s_gt[t] comes from the rank that owns gt(t) -> all-reduce(MAX) of Nt floats; every rank counts the
    better-scoring videos of its block -> all-reduce(SUM) of Nt int32.  No other collective; S is never gathered.

`compute` is the per-rank kernel backend (HipBackend below; the gloo/CPU tests inject an oracle-backed stand-in),
so the orchestration is exercised without a GPU.
"""
import torch
import torch.distributed as dist

from . import ops


def shard_bounds(n, world, rank):
    """Contiguous, balanced [lo, hi) -- the first n % world ranks get one extra row."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class HipBackend:
    """Per-rank compute on liblaff_hip.so."""

    def __init__(self, model, precision='fp16'):
        self.model, self.precision = model, precision
        ops.ctx_prepare_metrics(next(model.parameters()).device)     # scratch allocation outside any graph capture

    def embed_both(self, vis_feats, txt_feats):
        """Single-rank shortcut: both towers' FC projections in one grouped launch."""
        from .retrieval import embed
        self._emit_packed(True)
        try:
            return embed(self.model, vis_feats, txt_feats)
        finally:
            self._emit_packed(False)

    def txt_layer(self):
        return self.model.txt_net.attention_layer

    def vis_layer(self):
        net = self.model.vis_net
        return getattr(net, 'attention_layer', None) or getattr(net, 'vis_attention_layer', None)

    def embed_text(self, txt_feats):
        cap = dict(txt_feats)
        cap.setdefault('caption', None)
        self._emit_packed(True)
        try:
            return self.model.txt_net(cap)
        finally:
            self._emit_packed(False)

    def embed_video(self, vis_feats):
        vis = dict(vis_feats)
        frame_dict = {}
        if 'mask_tensor' in vis:
            frame_dict, vis = vis, {}
        self._emit_packed(True)
        try:
            return self.model.vis_net(vis, vis_frame_feat_dict_input=frame_dict)
        finally:
            self._emit_packed(False)

    def _attention_layers(self):
        return [getattr(net, name) for net in (self.model.vis_net, self.model.txt_net)
                for name in ('attention_layer', 'vis_attention_layer') if hasattr(net, name)]

    def _emit_packed(self, on):
        fused = self.precision in ('fp16', 'bf16')
        for layer in self._attention_layers():
            layer.emit_packed = self.precision if (on and fused and hasattr(layer, 'fuse_planes') and
                                                   type(layer).__name__ != 'JustAverage') else None

    def pack(self, E, layer=None):
        """GEMM operand of an embedding matrix: taken from the fuse launch when it emitted one, else a pack_rows pass."""
        p = getattr(layer, 'last_packed', None) if layer is not None else None
        if p is not None and p.N == E.shape[0]:
            layer.last_packed = None
            return p
        return ops.pack_rows(E, True, 1e-13, self.precision)

    def operand_from_gathered(self, bufs, rows, K, like):
        """Concatenated single-plane operand buffers of all ranks -> one Packed operand."""
        return ops.Packed(bufs, rows, K, like.precision, like.prescale)

    def sim(self, T, V, heads):
        return ops.sim_gemm(T, V, heads=heads)

    def row_dot_gt(self, T, V, gt, heads, col0):
        # the same launch clears the accumulator the fused count of sim_ranked adds into (no separate fill kernel)
        self._count = torch.empty((T.N,), dtype=torch.int32, device=T.buf.device)
        return ops.row_dot_gt(T, V, gt, heads, col0, zero_count=self._count)

    def sim_ranked(self, T, V, heads, gt, s_gt, col0, want_scores=True):
        """Score block + ground-truth rank counts in one GEMM launch (fused epilogue)."""
        count = getattr(self, '_count', None)
        if count is None or count.numel() != T.N:
            count = torch.zeros((T.N,), dtype=torch.int32, device=T.buf.device)
        self._count = None
        S = ops.sim_gemm(T, V, heads=heads, want_scores=want_scores, gt_col=gt, s_gt=s_gt, count=count, col0=col0)
        return S, count

    def finish(self, count, out_pinned=None):
        """ranks = count + 1 and the 7 metrics in one launch; returns (ranks, metrics or None when out_pinned is given)."""
        ranks = torch.empty_like(count)
        if out_pinned is not None:
            ops.rank_metrics_async(count, out_pinned, base=1, ranks_out=ranks)
            return ranks, None
        return ranks, ops.rank_metrics(count, base=1, ranks_out=ranks)

    def metrics(self, ranks):
        return ops.rank_metrics(ranks)

    def metrics_async(self, ranks, out_pinned):
        ops.rank_metrics_async(ranks, out_pinned)


class GraphRunner:
    """Runs each LOCAL phase of evaluate_sharded as its own captured HIP graph (first call: capture + replay, later calls:
    replay).  Collectives stay eager between the phases, so nothing RCCL-related is ever captured; what disappears is the
    per-step Python/ctypes issue cost of ~15 kernel launches (0.3-0.4 ms, several times the GPU time of a 1/8 shard)."""

    def __init__(self):
        self.graphs, self.outs = {}, {}

    def __call__(self, name, fn):
        if name not in self.graphs:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn()
            self.graphs[name], self.outs[name] = g, out
        self.graphs[name].replay()
        return self.outs[name]


def _eager(name, fn):
    return fn()


def evaluate_sharded(compute, vis_feats_local, txt_feats_local, gt, Nt, Nv, heads, group=None, want_metrics=True,
                     timer=None, want_scores=True, metrics_out=None, runner=None, state=None, force_collectives=False,
                     finish_tag=''):
    """One pass of the hot path on this rank's shards.  gt: (Nt,) int32 GLOBAL video column of every text (replicated).

    runner: None (eager) or a GraphRunner; state: dict that persists across steps (static collective buffers) -- required
    with a GraphRunner.  force_collectives runs the collectives even on a 1-rank group (used to exercise the N > 1 code
    path on a single GPU).  finish_tag names the captured 'finish' phase: callers that alternate between several metrics_out
    buffers (to keep a step in flight while the host reads the previous one) pass a different tag per buffer.
    Returns dict(S_local (Nt, v1-v0), col0, ranks (Nt,), metrics)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    comm = world > 1 or (force_collectives and dist.is_initialized())
    v0, v1 = shard_bounds(Nv, world, rank)
    mark = timer.mark if timer is not None else (lambda name: None)
    run = runner if runner is not None else _eager
    state = state if state is not None else {}
    if runner is not None and metrics_out is None and want_metrics:
        raise ValueError('a GraphRunner needs metrics_out (pinned buffer): the synchronising metrics call cannot be captured')
    sizes = [shard_bounds(Nt, world, r) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    with torch.no_grad():
        if not comm and hasattr(compute, 'embed_both'):
            def towers():
                vis_emb, txt_emb = compute.embed_both(vis_feats_local, txt_feats_local)
                return (txt_emb, compute.pack(txt_emb, compute.txt_layer()), vis_emb, compute.pack(vis_emb, compute.vis_layer()))
            txt_emb, T_all, vis_emb, V_local = run('towers', towers)
            mark('towers')
        else:
            def text_phase():
                txt_emb = compute.embed_text(txt_feats_local)
                T_local = compute.pack(txt_emb, compute.txt_layer()) if hasattr(compute, 'txt_layer') else compute.pack(txt_emb)
                if T_local.precision not in ('fp16', 'bf16'):
                    raise NotImplementedError("sharded evaluation gathers a single-plane 16-bit operand; precision '%s' "
                                              "is single-GPU only for now" % T_local.precision)
                row_bytes = T_local.K * 2
                send = T_local.buf[:T_local.N * row_bytes]
                if T_local.N != nmax:       # equal-sized contributions for all_gather_into_tensor
                    pad = torch.zeros(nmax * row_bytes, dtype=torch.uint8, device=send.device)
                    pad[:send.numel()] = send
                    send = pad
                return txt_emb, T_local, send.contiguous()
            txt_emb, T_local, send = run('text', text_phase)
            mark('txt_tower')
            row_bytes = T_local.K * 2
            work = None
            if comm:
                if 'gathered' not in state or state['gathered'].numel() != world * nmax * row_bytes:
                    state['gathered'] = torch.empty(world * nmax * row_bytes, dtype=torch.uint8, device=send.device)
                gathered = state['gathered']
                work = dist.all_gather_into_tensor(gathered, send, group=group, async_op=True)     # overlaps the video tower

            def video_phase():
                vis_emb = compute.embed_video(vis_feats_local)
                V = compute.pack(vis_emb, compute.vis_layer()) if hasattr(compute, 'vis_layer') else compute.pack(vis_emb)
                return vis_emb, V
            vis_emb, V_local = run('video', video_phase)
            mark('vis_tower')
            if work is None:
                T_all = T_local
            else:
                work.wait()
            if work is None:
                pass
            elif all(hi - lo == nmax for lo, hi in sizes):
                T_all = compute.operand_from_gathered(gathered, Nt, T_local.K, T_local)
            else:
                if runner is not None:
                    raise NotImplementedError('uneven text shards need a compaction copy that is not captured; use eager mode')
                parts = [gathered[r * nmax * row_bytes: r * nmax * row_bytes + (hi - lo) * row_bytes]
                         for r, (lo, hi) in enumerate(sizes)]
                T_all = compute.operand_from_gathered(torch.cat(parts), Nt, T_local.K, T_local)
            mark('all_gather_wait')
        # ground-truth score from the shard that owns the column, then one GEMM that writes S and counts ranks
        s_gt = run('s_gt', lambda: compute.row_dot_gt(T_all, V_local, gt, heads, v0))
        if comm:
            dist.all_reduce(s_gt, op=dist.ReduceOp.MAX, group=group)
        mark('s_gt')
        S_local, count = run('sim', lambda: compute.sim_ranked(T_all, V_local, heads, gt, s_gt, v0, want_scores))
        mark('sim_gemm')
        if comm:
            dist.all_reduce(count, op=dist.ReduceOp.SUM, group=group)

        metrics = None
        if hasattr(compute, 'finish') and (metrics_out is not None or want_metrics):
            # one launch: ranks = count + 1 and the metrics (no host sync when metrics_out is given)
            if metrics_out is not None:
                ranks = run('finish' + finish_tag, lambda: compute.finish(count, metrics_out)[0])
            else:
                ranks, metrics = compute.finish(count)
            mark('rank')
        else:
            def finish():
                ranks = count + 1
                if metrics_out is not None:
                    compute.metrics_async(ranks, metrics_out)     # no host sync: the caller reads metrics_out after one
                return ranks
            ranks = run('finish' + finish_tag, finish)
            mark('rank')
            if metrics_out is None and want_metrics:
                metrics = compute.metrics(ranks)
        mark('metrics')
    return {'S_local': S_local, 'col0': v0, 'ranks': ranks, 'metrics': metrics, 'vis_emb': vis_emb, 'txt_emb': txt_emb}


def evaluate_sharded_by_text(compute, vis_feats_local, txt_feats_local, gt, Nt, Nv, heads, group=None, want_metrics=True,
                             want_scores=True, metrics_out=None, runner=None, state=None, force_collectives=False,
                             finish_tag=''):
    """The OTHER decomposition the path admits (not the default: BASELINE.json asks for video-row shards with an all-gather of
    the text operand): rank g owns texts [t0, t1) end to end and the ROW block S[t0:t1, :]; videos are split for the embedding
    stage only and ONE all-gather of the 16-bit VIDEO operand (Nv x K: 10 MB at C4 instead of 41 MB for the texts) gives every
    rank all videos.  Every text then meets its ground-truth video locally, so neither the MAX all-reduce of s_gt nor the SUM
    all-reduce of the counts is needed; the ranks (Nt int32) are all-gathered for the replicated metrics.
    Returns dict(S_local (t1-t0, Nv), row0, ranks (Nt,), metrics)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    comm = world > 1 or (force_collectives and dist.is_initialized())
    t0, t1 = shard_bounds(Nt, world, rank)
    run = runner if runner is not None else _eager
    state = state if state is not None else {}
    if runner is not None and metrics_out is None and want_metrics:
        raise ValueError('a GraphRunner needs metrics_out (pinned buffer): the synchronising metrics call cannot be captured')
    vsizes = [shard_bounds(Nv, world, r) for r in range(world)]
    tsizes = [shard_bounds(Nt, world, r) for r in range(world)]
    vmax = max(hi - lo for lo, hi in vsizes)
    tmax = max(hi - lo for lo, hi in tsizes)
    even = all(hi - lo == vmax for lo, hi in vsizes) and all(hi - lo == tmax for lo, hi in tsizes)
    if runner is not None and comm and not even:
        raise NotImplementedError('uneven shards need compaction copies that are not captured; use eager mode')
    with torch.no_grad():
        def towers():
            if hasattr(compute, 'embed_both'):
                vis_emb, txt_emb = compute.embed_both(vis_feats_local, txt_feats_local)
            else:
                txt_emb, vis_emb = compute.embed_text(txt_feats_local), compute.embed_video(vis_feats_local)
            T_local = compute.pack(txt_emb, compute.txt_layer()) if hasattr(compute, 'txt_layer') else compute.pack(txt_emb)
            V_local = compute.pack(vis_emb, compute.vis_layer()) if hasattr(compute, 'vis_layer') else compute.pack(vis_emb)
            if V_local.precision not in ('fp16', 'bf16'):
                raise NotImplementedError("sharded evaluation gathers a single-plane 16-bit operand; precision '%s' is "
                                          'single-GPU only for now' % V_local.precision)
            row_bytes = V_local.K * 2
            send = V_local.buf[:V_local.N * row_bytes]
            if V_local.N != vmax:
                pad = torch.zeros(vmax * row_bytes, dtype=torch.uint8, device=send.device)
                pad[:send.numel()] = send
                send = pad
            return txt_emb, vis_emb, T_local, V_local, send.contiguous()
        txt_emb, vis_emb, T_local, V_local, send = run('towers_t', towers)
        row_bytes = V_local.K * 2
        if comm:
            if 'gathered_v' not in state or state['gathered_v'].numel() != world * vmax * row_bytes:
                state['gathered_v'] = torch.empty(world * vmax * row_bytes, dtype=torch.uint8, device=send.device)
            gathered = state['gathered_v']
            dist.all_gather_into_tensor(gathered, send, group=group)
            if even:
                V_all = compute.operand_from_gathered(gathered, Nv, V_local.K, V_local)
            else:
                parts = [gathered[r * vmax * row_bytes: r * vmax * row_bytes + (hi - lo) * row_bytes] for r, (lo, hi) in enumerate(vsizes)]
                V_all = compute.operand_from_gathered(torch.cat(parts), Nv, V_local.K, V_local)
        else:
            V_all = V_local
        gt_local = gt[t0:t1].contiguous()

        def rank_phase():
            s_gt = compute.row_dot_gt(T_local, V_all, gt_local, heads, 0)
            S_local, count = compute.sim_ranked(T_local, V_all, heads, gt_local, s_gt, 0, want_scores)
            mine = (count + 1).to(torch.int32)
            if mine.numel() != tmax:
                pad = torch.ones(tmax, dtype=torch.int32, device=mine.device)
                pad[:mine.numel()] = mine
                mine = pad
            return S_local, mine.contiguous()
        S_local, mine = run('rank_t', rank_phase)
        if comm:
            if 'gathered_r' not in state or state['gathered_r'].numel() != world * tmax:
                state['gathered_r'] = torch.empty(world * tmax, dtype=torch.int32, device=mine.device)
            all_ranks = state['gathered_r']
            dist.all_gather_into_tensor(all_ranks, mine, group=group)
            ranks = all_ranks if even else torch.cat([all_ranks[r * tmax: r * tmax + (hi - lo)] for r, (lo, hi) in enumerate(tsizes)])
        else:
            ranks = mine[:Nt]
        metrics = None
        if metrics_out is not None:
            run('finish_t' + finish_tag, lambda: compute.metrics_async(ranks, metrics_out))
        elif want_metrics:
            metrics = compute.metrics(ranks)
    return {'S_local': S_local, 'row0': t0, 'ranks': ranks, 'metrics': metrics, 'vis_emb': vis_emb, 'txt_emb': txt_emb}
