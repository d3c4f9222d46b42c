"""Agent-parallel When2com forward: one process per GPU, agents sharded across ranks, ONE exchange
step -- an all-gather of per-agent value maps V and keys K (RCCL over xGMI on MI355X; the
reference has no counterpart: its only multi-GPU scheme is nn.DataParallel batch splitting,
train.py:177).  SURVEY.md section 8e.

Rank r owns agents [r*n_loc, (r+1)*n_loc).  Encoders and policy net are per-image with shared
(replicated) eval-mode weights, so they need no communication; the communication graph needs
every agent's K and V (agent.py:1155) -- K travels as the PROJECTED key (Wq^T k | k.bq: Dq+1 = 33 floats per
agent-sample instead of 1024, the attention's Linear(query) is folded into the key head at pack time);
queries stay local; each rank fuses + decodes only its own
query agents.  xGMI is a point-to-point full mesh (7 links x ~153 GB/s per GPU): the V shard
(cfg 3: 2 MiB, cfg 4: 4 MiB bf16) is pushed once to each peer, ~14-27 us, and is issued
asynchronously as soon as the trunk's squeezer output exists so it overlaps the policy-net tail
(5 convs + 2 MLP heads).

``exchange_*`` are backend-agnostic (tested with gloo on CPU, world_size 2); the compute around
them is the HIP engine and needs a GPU.
"""
import os

import torch
import torch.distributed as dist

_SPARSE_FORCE = {"1": True, "0": False}.get(os.environ.get("W2C_SHARD_SPARSE", ""))        # see AgentParallelForward._sparse_pays
_SPARSE_MIN_BYTES = int(float(os.environ.get("W2C_SHARD_SPARSE_MIN_MB", "8")) * (1 << 20))


def _as_bytes_view(t):
    """all-gather the raw bytes (flat uint8 view): every backend (RCCL, gloo) moves uint8, not every
    backend registers bf16."""
    return t.view(torch.uint8).reshape(-1)


def exchange_start(local, group=None):
    """Begin all-gather of a contiguous per-rank shard [n_loc*B, ...] along dim 0.
    Returns (gathered tensor, work handle or None)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local, None
    if not local.is_contiguous():
        raise ValueError("exchange: shard must be contiguous")
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(_as_bytes_view(out), _as_bytes_view(local), group=group, async_op=True)
    return out, work


def exchange_wait(work):
    if work is not None:
        work.wait()          # on NCCL/RCCL: makes the current stream wait for the collective


def plan_sparse_exchange(need, B, N, world, rank):
    """Handshake plan (the paper's point, agent.py:1036-1078): after keys and queries (33 + 32 floats per
    agent-sample) have been exchanged every rank holds the same communication graph, so it also knows which value maps
    V[b, k] any query agent of which rank will actually use (coef != 0: P > 0.2 in 'activated', the argmax in
    'argmax_test').  need: bool [B, N_keys, N_queries] on the host.  Rows are agent-major (agent*B + b).
    Returns (send_rows[dst] -> LOCAL row indices of v_local, recv_rows[src] -> GLOBAL row indices of v_all), both sorted,
    identical on both ends of every pair by construction."""
    n_loc = N // world
    need = torch.as_tensor(need, dtype=torch.bool).reshape(B, N, world, n_loc).any(dim=3)       # [B, N_keys, dst rank]
    send_rows, recv_rows = [], []
    for peer in range(world):
        if peer == rank:
            send_rows.append([])
            recv_rows.append([])
            continue
        mine = need[:, rank * n_loc:(rank + 1) * n_loc, peer]                                   # [B, n_loc]: my keys the peer needs
        send_rows.append([kl * B + b for kl in range(n_loc) for b in range(B) if bool(mine[b, kl])])
        theirs = need[:, peer * n_loc:(peer + 1) * n_loc, rank]                                 # the peer's keys I need
        recv_rows.append([(peer * n_loc + kl) * B + b for kl in range(n_loc) for b in range(B) if bool(theirs[b, kl])])
    return send_rows, recv_rows


def sparse_exchange(v_local, need, B, N, group=None):
    """Transfer ONLY the value maps the communication graph uses (one all-to-all with per-pair sizes) instead of
    all-gathering every agent's map.  v_local [n_loc*B, ...] contiguous; returns (v_all [N*B, ...] with this rank's rows,
    the received rows, zeros elsewhere -- those rows have fusion weight 0 --, maps_received, maps_dense) where maps_dense =
    (world-1)*n_loc*B is what the all-gather would have pulled in."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_loc = N // world
    v_all = torch.zeros((N * B,) + tuple(v_local.shape[1:]), dtype=v_local.dtype, device=v_local.device)
    v_all[rank * n_loc * B:(rank + 1) * n_loc * B] = v_local
    if world == 1:
        return v_all, 0, 0
    send_rows, recv_rows = plan_sparse_exchange(need, B, N, world, rank)
    row_bytes = v_local[0].numel() * v_local.element_size()
    flat_send = [r for rows in send_rows for r in rows]
    flat_recv = [r for rows in recv_rows for r in rows]
    dev = v_local.device
    send_buf = (v_local.index_select(0, torch.tensor(flat_send, dtype=torch.long, device=dev)) if flat_send
                else v_local.new_empty((0,) + tuple(v_local.shape[1:])))
    recv_buf = v_local.new_empty((len(flat_recv),) + tuple(v_local.shape[1:]))
    dist.all_to_all_single(_as_bytes_view(recv_buf), _as_bytes_view(send_buf.contiguous()),
                           output_split_sizes=[len(r) * row_bytes for r in recv_rows],
                           input_split_sizes=[len(r) * row_bytes for r in send_rows], group=group)
    if flat_recv:
        v_all.index_copy_(0, torch.tensor(flat_recv, dtype=torch.long, device=dev), recv_buf)
    return v_all, len(flat_recv), (world - 1) * n_loc * B


def shard_agents(agent_num, world, rank):
    if agent_num % world != 0:
        raise ValueError("agent_num %d is not divisible by world size %d" % (agent_num, world))
    n_loc = agent_num // world
    return rank * n_loc, n_loc


def _gather_inplace(buf, rank, rows, group=None):
    """all-gather where every rank's shard already sits in its own rows [rank*rows, (rank+1)*rows) of `buf` (the
    producing kernels wrote it there: no staging copy).  RCCL does this in place; other backends (gloo in the tests)
    get a private copy of the shard as the send buffer.  Returns the async work handle."""
    mine = buf[rank * rows:(rank + 1) * rows]
    if dist.get_backend(group) != "nccl":
        mine = mine.clone()
    return dist.all_gather_into_tensor(_as_bytes_view(buf), _as_bytes_view(mine), group=group, async_op=True)


class _ShardState:
    """Static buffers + captured segments of one (engine, input shape) pair on one rank."""

    def __init__(self, eng, x, n_loc, N, q_lo):
        B, _, H, W = x.shape
        dev = x.device
        h, w = H // 32, W // 32
        bf16 = torch.bfloat16
        self.B, self.N, self.n_loc, self.q_lo = B, N, n_loc, q_lo
        self.s0 = torch.empty((n_loc * B, H // 4, W // 4, 64 * eng.trunk.G), dtype=bf16, device=dev)   # pooled stem output
        # What crosses the wire per agent-sample is U = the decoder's first conv of the value map (f32, 256 channels: the same 1 KiB
        # per pixel as the bf16 512-channel V it replaces; engine.DecoderPlan.value_maps -- conv0 is linear before its bias, and so is
        # the fusion).  MIMOcomWho's second map U_own (the conv of V[q] with the other half of conv0's filters, used by the agent's OWN
        # decode only) never leaves the rank (round 5; round 4 gathered [U | U_own]: twice the xGMI bytes the model needs).
        d = eng.decoder
        ucs = d.c_hidden
        self.v_loc = torch.empty((n_loc * B, h, w, eng.feat), dtype=bf16, device=dev)   # local value maps (the value trunk's squeezer)
        self.v_all = torch.zeros((N * B, h, w, ucs), dtype=torch.float32, device=dev)   # every agent's U map
        self.u_own = torch.zeros((n_loc * B, h, w, ucs), dtype=torch.float32, device=dev) if d.own_off >= 0 else None
        self.pol = torch.empty((n_loc * B, h, w, eng.feat), dtype=bf16, device=dev)     # local policy-encoder map
        dq = eng.wq.shape[1]
        self.k_all = torch.zeros((N * B, dq + 1), dtype=torch.float32, device=dev)      # projected keys of every agent
        self.q_loc = torch.empty((n_loc * B, dq), dtype=torch.float32, device=dev) if eng.has_query else None
        lo, hi = q_lo * B, (q_lo + n_loc) * B
        self.v_slot = self.v_all[lo:hi]
        self.k_slot = self.k_all[lo:hi]
        self.pol_y = None               # policy conv5's output (segment A, static under graph replay): the heads' input
        self.graphs = {}
        self.out = {}

    def run(self, name, fn, use_graph):
        """Run segment `name` (fn() -> tuple of tensors written into static buffers), eagerly or by replaying its recorded
        program (ops.record_program: single-branch HIP graphs on this rank's lanes; recorded on first use, after two warm-up runs)."""
        if not use_graph:
            self.out[name] = fn()
            return self.out[name]
        prog = self.graphs.get(name)
        if prog is None:
            from . import ops
            prog = self.graphs[name] = ops.record_program(self.v_all.device, fn, warmup=2)
        self.out[name] = prog.replay()
        return self.out[name]


class AgentParallelForward:
    """Callable mirroring MIMOcom.forward for a rank's local agents.

    inputs_local: f32 [B, 3*n_loc, H, W] (this rank's agents' frames).
    Returns (pred [n_loc*B, n_cls, H, W], prob [B, N, n_loc], action [B, n_loc], nnz [B])
    for the local query agents; N = global agent count.

    Per step and rank (dense exchange: 'softmax', and the thresholded modes below _sparse_pays' limit) -- _dense_step:
        stem (reads the caller's frames through a pointer slot) -> layer1 (both trunks) -> layer2.0's front -> fork
          lane 1, value chain : layer2..4, squeezer, decoder conv0 on the local value maps (U, by linearity: engine.DecoderPlan.value_maps,
                                written straight into this rank's rows of the gather buffer), in-place RCCL all-gather of U -- issued from
                                the value chain, so it travels under the policy chain's tail
          lane 0, policy chain: layer2..4, squeezer, policy conv1..5, heads (projected keys into this rank's rows), all-gather of K
        join -> graph columns of the local queries + fusion of the U maps + bias + ReLU -> decoder's last conv -> x32 upsample into the
        caller-owned logits (pointer slot) -> packed prob / action / nnz into the caller-owned copy.
    With model.use_hip_graph the step is ONE recorded program (ops.record_program): ~45 kernel launches become a handful of
    single-branch HIP graphs on the rank's two lanes; the two collectives are issued by the host between them at every replay (round 6:
    rounds 4-5 captured them INTO a graph, which needed a proof that ProcessGroupNCCL's watchdog list was empty at capture time -- a
    private torch API -- and a graph with parallel branches; an eager collective between two graph launches costs ~20 us of host time
    per step and none of that).  Sparse exchange and the fp8 trunk: stem | segment A | collectives | B | C, each segment a program."""

    def __init__(self, model, group=None):
        from . import engine as _engine
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.q_lo, self.n_loc = shard_agents(model.agent_num, self.world, self.rank)
        self._engine_cls = _engine.CommEngine
        self.last_exchange = None
        # force_sharded: take the multi-rank code path (the recorded program + in-place RCCL all-gathers) even with ONE rank -- the
        # only way to execute the RCCL calls between the graph launches on a single-GPU box (tests, bench --force-sharded); results
        # equal forward_local's bit for bit
        self.force_sharded = os.environ.get("W2C_FORCE_SHARDED", "0") == "1"
        self.launch_form = "eager"

    def _state(self, eng, x):
        cache = eng.__dict__.setdefault("_shard_states", {})          # dies with the engine (load_state_dict / .to() / train())
        key = (tuple(x.shape), self.world, self.rank)
        st = cache.get(key)
        if st is None:
            st = _ShardState(eng, x, self.n_loc, self.model.agent_num, self.q_lo)
            cache[key] = st
        return st

    def __call__(self, inputs_local, inference="softmax"):
        from . import ops
        model = self.model
        if model.training:
            raise RuntimeError("agent-parallel forward is the eval (HIP) path; call model.eval()")
        eng = model._engine_for(inputs_local, self._engine_cls)   # a dict lookup; never cached here (load_state_dict /
                                                                  # .to() / train() drop the model's packed weights)
        B = inputs_local.shape[0]
        N = model.agent_num
        use_graph = bool(getattr(model, "use_hip_graph", False))
        with torch.no_grad():
            x = inputs_local.contiguous().float()
            if self.world == 1 and not (self.force_sharded and dist.is_initialized()):
                return eng.forward_local(x, B, N, inference, use_graph=use_graph)
            if (inference == "softmax" or not self._sparse_pays(eng, x)) and not eng.trunk.n8 and dist.is_initialized():
                return self._dense_step(eng, x, B, N, inference, use_graph)
            self.launch_form = "3 program segments + eager collectives" if use_graph else "eager"
            st = self.encode_local(eng, x, use_graph)
            if inference != "softmax":
                return self._sparse(eng, st, inference, use_graph)
            v_work = _gather_inplace(st.v_all, self.rank, self.n_loc * B, self.group)
            st.run("B", lambda: self._segment_b(eng, st), use_graph)   # under the V gather
            k_work = _gather_inplace(st.k_all, self.rank, self.n_loc * B, self.group)
            exchange_wait(v_work)
            exchange_wait(k_work)
            low, prob, action, nnz = st.run(
                "C:softmax", lambda: eng.graph_and_low(st.v_all, st.k_all, st.q_loc, B, N, self.q_lo, self.n_loc, "softmax", u_own=st.u_own),
                use_graph)
            pred = ops.upsample_bilinear32(low, eng.n_classes)
        if use_graph:
            prob, action, nnz = prob.clone(), action.clone(), nnz.clone()
        return pred, prob, action, nnz

    def _sparse_pays(self, eng, x):
        """'activated' / 'argmax_test' across ranks: handshake-ordered sparse exchange (self._sparse: only the maps with a non-zero
        fusion weight cross the links, but the transfer sizes are host arguments of the all-to-all -- one host round trip, eager
        launches behind it) or the dense all-gather inside the one-graph step (every map crosses, nothing leaves the device)?
        The dense step costs the unused maps' bytes, the sparse one ~0.1 ms of host round trip + eager launches: dense below
        W2C_SHARD_SPARSE_MIN_MB (default 8) MiB of maps received per rank and step -- every BASELINE config (cfg 2: 4.2 MB) -- sparse
        above.  W2C_SHARD_SPARSE=1 / 0 forces one."""
        if _SPARSE_FORCE is not None:
            return _SPARSE_FORCE
        B, _, H, W = x.shape
        ucs = 256                                           # U alone crosses the links (MIMOcomWho's U_own stays on the rank)
        recv = (self.world - 1) * self.n_loc * B * (H // 32) * (W // 32) * ucs * 4
        return recv >= _SPARSE_MIN_BYTES

    def _dense_step(self, eng, x, B, N, inference, use_graph):
        """The whole sharded step of a rank with the dense exchange (see the class docstring), run eagerly or replayed from ONE recorded
        program.  'activated' / 'argmax_test' take it when the dense exchange is the cheaper one (_sparse_pays): the communication-graph
        kernel zeroes the coefficients of the unused maps either way, so the results equal the sparse path's bit for bit."""
        from . import ops
        st = self._state(eng, x)
        n_loc, q_lo = self.n_loc, self.q_lo
        dev = x.device
        H, W = x.shape[2], x.shape[3]
        rows = n_loc * B
        out = torch.empty((n_loc * B, eng.n_classes, H, W), dtype=torch.float32, device=dev)

        def step(io):
            # io: this run's caller-owned tensors; the host-issued regions (stem, join) read them at every replay as plain arguments
            L = ops.lanes(dev)
            works = []
            L.eager(lambda: eng.trunk.stem(io["x"], n_loc, out=st.s0))      # host-issued at every replay (engine.TrunkPlan.after_stem says why)

            def value_tail(v):
                # (lane 1) U first, K behind it: the process group runs its collectives in issue order on ONE internal stream, so the K
                # gather (which waits for the end of the policy tail) must not be queued ahead of the U gather, or U would not travel
                # under the policy tail.  after_stem hands the value chain its launches first, the policy chain's second.
                u = eng.value_maps(v, out=st.v_slot, out_own=st.u_own)
                L.eager(lambda: works.append(_gather_inplace(st.v_all, self.rank, rows, self.group)))
                return u

            def policy_tail(pol):
                # (lane 0)
                y = eng.policy_convs(pol, ch_off=0)
                eng.policy_heads(y, outs=(st.k_slot, st.q_loc))
                L.eager(lambda: works.append(_gather_inplace(st.k_all, self.rank, rows, self.group)))
                return y

            eng.trunk.after_stem(st.s0, squeezer_out=[st.v_loc, st.pol], policy_next=(policy_tail, lambda y: y), value_next=value_tail)

            def wait_all():
                while works:
                    exchange_wait(works.pop(0))
            def join():
                wait_all()
                low, prob, action, nnz = eng.graph_and_low(st.v_all, st.k_all, st.q_loc, B, N, q_lo, n_loc, inference, pack2=io["pack"], u_own=st.u_own)
                ops.upsample_bilinear32(low, eng.n_classes, out=io["out"])
            L.eager(join)
            return eng._last_pack

        dense = (self.world - 1) * n_loc * B
        self.last_exchange = (dense, dense)
        if not use_graph:
            self.launch_form = "eager"
            packc = ops.graph_outputs(dev, B, N, n_loc)[0]
            step(dict(x=x, out=out, pack=packc))
            prob, action, nnz = ops.carve_graph_outputs(packc, B, N, n_loc)
            return out, prob, action, nnz
        ent = st.graphs.get("whole:" + inference)
        if ent is None:
            io = dict(x=x, out=out, pack=ops.graph_outputs(dev, B, N, n_loc)[0])
            program = ops.record_program(dev, lambda: step(io), warmup=2)
            ent = st.graphs["whole:" + inference] = (program, io)
        program, io = ent
        packc = torch.empty_like(program.result)
        io.update(x=x, out=out, pack=packc)
        try:
            program.replay()
        finally:
            io.update(x=None, out=None, pack=None)
        self.launch_form = "one program: %d single-branch hip-graphs + %d host-issued steps incl. the RCCL all-gathers" % (program.n_graphs, program.n_calls)
        prob, action, nnz = ops.carve_graph_outputs(packc, B, N, n_loc)
        return out, prob, action, nnz

    def encode_local(self, eng, x, use_graph=False):
        """stem + segment A on this rank's frames: the U maps of the local agents (decoder conv0 of their value maps) land in
        st.v_slot (= their rows of st.v_all), the policy-encoder map in st.pol.  Returns the shard state."""
        st = self._state(eng, x)
        eng.trunk.stem(x, self.n_loc, out=st.s0)
        if eng.trunk.n8 and eng.trunk.fp8 is None:
            # fp8 trunk: every rank must quantise with the SAME activation scales, or a shard would round differently from the
            # unsharded batch -- the calibration pass's amax vector is all-reduced (MAX) once, before anything is captured
            def _max_over_ranks(amax):
                if dist.is_initialized() and self.world > 1:
                    dist.all_reduce(amax, op=dist.ReduceOp.MAX, group=self.group)
                return amax
            eng.trunk.calibrate(st.s0, reduce_amax=_max_over_ranks)
        if eng.trunk.n8:
            st.run("A", lambda: (eng.trunk.after_stem(st.s0, squeezer_out=[st.v_loc, st.pol],
                                                      value_next=lambda v: eng.value_maps(v, out=st.v_slot, out_own=st.u_own))[2],), use_graph)
            st.pol_y = None
        else:
            # as in the one-GPU forward, policy conv1..5 ride the policy chain's stream beside the value chain (they were the whole of
            # the sharded path's extra 0.09 ms per step when segment B ran them after the join); segment B keeps the heads
            a = st.run("A", lambda: (eng.trunk.after_stem(st.s0, squeezer_out=[st.v_loc, st.pol],
                                                          policy_next=(lambda pol: eng.policy_convs(pol, ch_off=0), lambda y: y),
                                                          value_next=lambda v: eng.value_maps(v, out=st.v_slot, out_own=st.u_own))[1],),
                       use_graph)
            st.pol_y = a[0]
        return st

    @staticmethod
    def _segment_b(eng, st):
        if st.pol_y is None:
            return eng.policy_tail(st.pol, ch_off=0, outs=(st.k_slot, st.q_loc))
        return eng.policy_heads(st.pol_y, outs=(st.k_slot, st.q_loc))

    def _sparse(self, eng, st, inference, use_graph=False):
        """'activated' / 'argmax_test' across ranks, handshake-ordered: tiny exchange of projected keys
        and queries -> every rank evaluates the whole graph -> only the value maps with a non-zero fusion weight cross
        xGMI.  self.last_exchange = (maps received, maps an all-gather would have received)."""
        from . import ops
        B, N = st.B, st.N
        st.run("B", lambda: self._segment_b(eng, st), use_graph)
        k_work = _gather_inplace(st.k_all, self.rank, self.n_loc * B, self.group) if dist.is_initialized() else None
        q_all, q_work = (exchange_start(st.q_loc, self.group) if st.q_loc is not None else (None, None))
        exchange_wait(k_work)
        exchange_wait(q_work)
        _, coef_full, _, _ = ops.comm_graph_projected(q_all, st.k_all, B, N, eng.who, inference)   # [B, N, N], same on every rank
        need = (coef_full != 0).cpu()                                          # the handshake's one host round trip
        v_all, got, dense = sparse_exchange(st.v_slot, need, B, N, self.group)
        self.last_exchange = (got, dense)
        pred, prob, action, nnz, _ = eng.graph_and_decode(v_all, st.k_all, st.q_loc, B, N, self.q_lo, self.n_loc, inference, u_own=st.u_own)
        return pred, prob, action, nnz


# ---------------------------------------------------------------------------------------------------------------------------------
# Agent-sharded TRAINING step (round 4; SURVEY 8f rank 3 + 8e: "train-mode BN couples agents across ranks").  The reference trains
# MIMOcom with every agent on one device (trainer.py:669-673); here rank r holds the frames of its own agents and
#   * every train-mode BatchNorm takes its statistics over ALL ranks' pixels (train_ops.set_sync_bn: the reference normalises over the
#     agent-concatenated batch, agent.py:1108-1111) -- forward sums and backward sums all-reduced between partial sums and finalize;
#   * value maps and keys are all-gathered DIFFERENTIABLY (backward: the gradient contributions of every rank's loss, summed, this
#     rank's rows), queries stay local, each rank fuses + decodes + takes the loss of its own query agents;
#   * parameter gradients are all-reduced (SUM) in one flattened bucket; every rank backpropagates its share of the GLOBAL loss
#     (_sharded_loss: local sum / global denominator for the size_average cross entropy, local loss / world otherwise).
# The result equals the unsharded step up to summation order (tests/test_parallel_gpu.py: loss and gradients against the one-GPU step).
class _AllGatherRows(torch.autograd.Function):
    """y = concat over the ranks (rank order) of x along dim 0; dy -> sum over the ranks of dy, this rank's rows."""

    @staticmethod
    def forward(ctx, x, group):
        world = dist.get_world_size(group)
        ctx.group, ctx.rows, ctx.rank = group, x.shape[0], dist.get_rank(group)
        x = x.contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(_as_bytes_view(out), _as_bytes_view(x), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g[ctx.rank * ctx.rows:(ctx.rank + 1) * ctx.rows], None


def agent_parallel_train_forward(model, inputs_local, group=None):
    """Train-mode forward of MIMOcom / MIMOcomWho for this rank's agents (inputs_local f32 [B, 3*n_loc, H, W]) ->
    (pred [n_loc*B, n_cls, H, W] with autograd history, prob [B, N, n_loc]).  Call under train_ops.set_sync_bn(True, group)."""
    from . import train_ops
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    N = model.agent_num
    q_lo, n_loc = shard_agents(N, world, rank)
    B = inputs_local.shape[0]
    unified = torch.cat([inputs_local[:, 3 * i:3 * i + 3] for i in range(n_loc)], 0)
    if inputs_local.is_cuda and train_ops.bf16_activations():
        unified = unified.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    feat = model.u_encoder(unified).float()
    qk = model.query_key_net(unified)
    keys = model.key_net(qk)
    feat_all = _AllGatherRows.apply(feat, group)                       # agent-major rows of every agent (rank order = agent order)
    keys_all = _AllGatherRows.apply(keys, group)
    val_mat = torch.stack([feat_all[B * i:B * (i + 1)] for i in range(N)], 1)
    key_mat = torch.stack([keys_all[B * i:B * (i + 1)] for i in range(N)], 1)
    if model.has_query:
        qs = model.query_net(qk)
        query_mat = torch.stack([qs[B * i:B * (i + 1)] for i in range(n_loc)], 1)
    else:
        query_mat = torch.ones(B, n_loc, model.query_size, device=inputs_local.device)
    scores = torch.bmm(key_mat, model.attention_net.linear(query_mat).transpose(2, 1))        # [B, N keys, n_loc queries]
    who = bool(getattr(model, "_who", False))
    if who:
        mask = torch.zeros(N, n_loc, dtype=torch.bool, device=inputs_local.device)
        mask[torch.arange(q_lo, q_lo + n_loc), torch.arange(n_loc)] = True
        scores = scores.masked_fill(mask.unsqueeze(0), float("-inf"))
    prob = torch.softmax(scores, dim=1)
    fused = torch.einsum("bkq,bkchw->bqchw", prob, val_mat)
    if who:
        fused = torch.cat((fused, val_mat[:, q_lo:q_lo + n_loc]), dim=2)
    pred = model.decoder(torch.cat([fused[:, i] for i in range(n_loc)], 0))
    return pred, prob


def _sharded_loss(loss_fn, pred, labels_local, group):
    """-> (the term THIS rank backpropagates, the global loss as a detached float tensor).
    The pixel-wise cross entropy with size_average (loss.py:5-18, the reference's default) is sum(w_t nll) / sum(w_t) over the kept
    pixels of the WHOLE batch; with ignore_index pixels or class weights the ranks' denominators differ, so the mean of the ranks' means
    is not it.  For cross_entropy2d (plain or functools.partial of it) each rank takes its local SUM, the denominators
    are all-reduced, and the rank backpropagates local_sum / global_denominator.  Any other loss: the mean of the ranks' values
    (exact for losses that are a mean over images of per-image terms with equal images per rank, e.g. bootstrapped_cross_entropy2d)."""
    import functools
    from . import loss as _loss
    world = dist.get_world_size(group)
    base, kw = loss_fn, {}
    while isinstance(base, functools.partial):
        kw = {**base.keywords, **kw}
        if base.args:
            base = None
            break
        base = base.func
    if base is _loss.cross_entropy2d and kw.get("size_average", True) and set(kw) <= {"weight", "size_average"}:
        weight = kw.get("weight")
        n_cls = pred.shape[1]
        local_sum = _loss.cross_entropy2d(pred, labels_local, weight=weight, size_average=False)
        lab = labels_local.reshape(-1).long()
        valid = (lab >= 0) & (lab < n_cls) & (lab != _loss.IGNORE_INDEX)
        if weight is None:
            denom = valid.sum().double().reshape(1)
        else:
            wv = weight.to(device=lab.device, dtype=torch.float64)
            denom = (wv[lab.clamp(0, n_cls - 1)] * valid.double()).sum().reshape(1)
        tot = torch.cat([denom, local_sum.detach().double().reshape(1)])
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
        d = tot[0].clamp_min(1e-30)
        return local_sum / d.float(), (tot[1] / d).float()
    if base is not None and getattr(base, "__name__", "") in ("cross_entropy2d", "multi_scale_cross_entropy2d"):
        import warnings
        warnings.warn("agent-parallel training: %s with these keywords is averaged as the mean of the ranks' losses, which differs from the "
                      "unsharded loss when ignored pixels or class weights differ across ranks" % base.__name__, stacklevel=2)
    local = loss_fn(pred, labels_local)
    g = local.detach().clone().float().reshape(1)
    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return local / world, (g / world).reshape(())


def agent_parallel_train_step(model, optimizer, loss_fn, inputs_local, labels_local, group=None):
    """One agent-sharded training step (see the block comment above) -> the GLOBAL loss as a float tensor (for cross_entropy2d: the
    size_average loss over every rank's pixels with the global denominator, see _sharded_loss).
    labels_local: the labels of this rank's agents, agent-major [n_loc*B, H, W].  The caller's sync-BN setting is restored on exit;
    a train-mode BatchNorm that cannot take the synchronised HIP path raises (train_ops.bn_act) instead of normalising locally."""
    from . import loss as _loss
    from . import train_ops
    prev = train_ops.set_sync_bn(True, group)
    # every rank calls the loss the same number of times here, so the out-of-range-label check may be collective: a bad label on ONE
    # rank raises on ALL of them instead of leaving its peers blocked in the gradient all-reduce
    prev_chk = _loss.set_label_check_collective(True, group)
    try:
        optimizer.zero_grad(set_to_none=True)
        pred, _ = agent_parallel_train_forward(model, inputs_local, group)
        term, global_loss = _sharded_loss(loss_fn, pred, labels_local, group)
        term.backward()
    finally:
        _loss.restore_label_check_collective(prev_chk)
        train_ops.restore_sync_bn(prev)
    params = [p for p in model.parameters() if p.grad is not None]
    if params:
        flat = torch.cat([p.grad.reshape(-1).float() for p in params])         # one bucket: one collective
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for p in params:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
    optimizer.step()
    return global_loss
