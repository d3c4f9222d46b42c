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
import torch
import torch.distributed as dist


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


class AgentParallelForward:
    """Callable mirroring MIMOcom.forward for a rank's local agents.

    inputs_local: f32 [B, 3*n_loc, H, W] (this rank's agents' frames).
    Returns (pred [n_loc*B, n_cls, H, W], prob [B, N, n_loc], action [B, n_loc], nnz [B])
    for the local query agents; N = global agent count."""

    def __init__(self, model, group=None):
        from . import engine as _engine
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.q_lo, self.n_loc = shard_agents(model.agent_num, self.world, self.rank)
        self._engine_cls = _engine.CommEngine

    def __call__(self, inputs_local, inference="softmax"):
        model = self.model
        if model.training:
            raise RuntimeError("agent-parallel forward is the eval (HIP) path; call model.eval()")
        eng = model._engine_for(inputs_local, self._engine_cls)   # a dict lookup; never cached here (load_state_dict /
                                                                  # .to() / train() drop the model's packed weights)
        B = inputs_local.shape[0]
        N = model.agent_num
        with torch.no_grad():
            x = inputs_local.contiguous().float()
            if self.world == 1:
                return eng.forward_local(x, B, N, inference, use_graph=getattr(model, "use_hip_graph", False))
            sq = eng.trunk.run(x, self.n_loc)                                  # [n_loc*B,h,w,1024]
            if inference != "softmax":
                return self._sparse(eng, sq, B, N, inference)
            v_loc = sq[..., :eng.feat].contiguous() if self.world > 1 else None
            v_all, v_work = exchange_start(v_loc, self.group) if self.world > 1 else (None, None)
            keys, querys = eng.policy_tail(sq)                                 # overlaps the V all-gather
            if self.world > 1:
                k_all, k_work = exchange_start(keys, self.group)
                exchange_wait(v_work)
                exchange_wait(k_work)
                v_src, v_ch = v_all, eng.feat
            else:
                k_all, v_src = keys, sq
            pred, prob, action, nnz, _ = eng.graph_and_decode(v_src, k_all, querys, B, N, self.q_lo, self.n_loc,
                                                              inference)
        return pred, prob, action, nnz

    def _sparse(self, eng, sq, B, N, inference):
        """'activated' / 'argmax_test' across ranks, handshake-ordered: tiny exchange of projected keys
        and queries -> every rank evaluates the whole graph -> only the value maps with a non-zero fusion weight cross
        xGMI.  self.last_exchange = (maps received, maps an all-gather would have received)."""
        from . import ops
        keys, querys = eng.policy_tail(sq)
        k_all, k_work = exchange_start(keys, self.group)
        q_all, q_work = (exchange_start(querys, self.group) if querys is not None else (None, None))
        exchange_wait(k_work)
        exchange_wait(q_work)
        _, coef_full, _, _ = ops.comm_graph_projected(q_all, k_all, B, N, eng.who, inference)     # [B, N, N], same on every rank
        need = (coef_full != 0).cpu()                                          # the handshake's one host round trip
        v_all, got, dense = sparse_exchange(sq[..., :eng.feat].contiguous(), need, B, N, self.group)
        self.last_exchange = (got, dense)
        pred, prob, action, nnz, _ = eng.graph_and_decode(v_all, k_all, querys, B, N, self.q_lo, self.n_loc, inference)
        return pred, prob, action, nnz
