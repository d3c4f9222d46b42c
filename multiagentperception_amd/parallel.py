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
        self.eng = None

    def __call__(self, inputs_local, inference="softmax"):
        model = self.model
        if model.training:
            raise RuntimeError("agent-parallel forward is the eval (HIP) path; call model.eval()")
        if self.eng is None:
            self.eng = model._engine_for(inputs_local, self._engine_cls)
        eng = self.eng
        B = inputs_local.shape[0]
        N = model.agent_num
        with torch.no_grad():
            x = inputs_local.contiguous().float()
            if self.world == 1:
                return eng.forward_local(x, B, N, inference, use_graph=getattr(model, "use_hip_graph", False))
            sq = eng.trunk.run(x, self.n_loc)                                  # [n_loc*B,h,w,1024]
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
