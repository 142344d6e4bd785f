"""Training step of the decoder on the HIP kernels: forward with stashed activations, backward, ZeRO-2-style sharded AdamW.

Reference: `forward(inference=False)` builds the loss (ufvideo/model/videorefer_qwen2.py:198-352), torch autograd over HF
Qwen2 (modeling_qwen2.py) differentiates it, and DeepSpeed ZeRO-2 + AdamW applies it (ufvideo/train.py:749,
scripts/zero2.json: gradients reduce-scattered, optimizer states partitioned, updated parameters all-gathered).

What is built here (first slice): the causal-LM cross-entropy objective through the Qwen2
decoder stack -- lm_head, final norm, every decoder layer, and the embed_tokens rows of the text positions -- i.e. the
61 % of the training FLOPs that sit in the LLM.  The gradient with respect to the visual tokens is returned
(`d_inputs_embeds`) for a projector backward that is not built yet; the vision tower is frozen in the reference
(encoder.py:122,134); the mask-loss branch (SAM2 mask decoder) has forward values only.

MI355X layout: 288 GB per GPU holds bf16 weights + transposed copies + fp32 gradients + fp32 master/m/v of a 7B decoder on
ONE GPU, and every layer's activations (0.4 GB per layer at S = 2400) are kept instead of re-computed (the reference turns
gradient checkpointing on for 80 GB parts).  Parameters live in one flat bf16 buffer per layer ("bucket") with a flat fp32
gradient buffer beside it, so the data-parallel exchange is one reduce-scatter per bucket, issued when that layer's
backward has finished and overlapped with the next layer's, and one all-gather of the updated bf16 shard.
"""
import math
import os

import torch
import torch.distributed as dist

from . import ops, _lib


def _ru(x, m):
    return (x + m - 1) // m * m


def reduce_scatter_mean(out_shard, full, group=None):
    """out_shard = this rank's slice of mean-over-ranks(full).  RCCL: one reduce-scatter; gloo (CPU tests) has none, so
    all-reduce and keep the slice."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = out_shard.numel()
    assert full.numel() == n * world
    if dist.get_backend(group) == "gloo":
        tmp = full.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
        out_shard.copy_(tmp[rank * n:(rank + 1) * n])
    else:
        dist.reduce_scatter_tensor(out_shard, full, op=dist.ReduceOp.SUM, group=group)
    out_shard.mul_(1.0 / world)
    return out_shard


def all_gather_shards(full, group=None):
    """full[rank's slice] holds this rank's updated shard; fill in everybody else's"""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = full.numel() // world
    mine = full[rank * n:(rank + 1) * n].clone()
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        full.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(full, mine, group=group)
    return full


class _Bucket:
    """Parameters that are exchanged and updated together: flat bf16 working copy `wb` (or fp32 `wf` for the small
    fp32 parameters), flat fp32 gradient `g`, and this rank's shard of master / m / v."""

    def __init__(self, names_shapes, device, world, rank, dtype, decay, trainable=True, keep_grad=False):
        self.trainable = trainable
        self.entries = []
        off = 0
        for name, shape in names_shapes:
            n = int(math.prod(shape))
            self.entries.append((name, tuple(shape), off, n))
            off = _ru(off + n, 64)                                # keep every view 128-byte aligned
        self.n = _ru(off, 64 * world)
        self.world, self.rank, self.decay = world, rank, decay
        self.dtype = dtype
        self.w = torch.zeros((self.n,), device=device, dtype=dtype)
        self.g = torch.zeros((self.n,), device=device, dtype=torch.float32) if (trainable or keep_grad) else None
        self.shard = self.n // world
        self.master = None
        self.work = None                                          # pending reduce-scatter

    def view(self, buf, name):
        for nm, shape, off, n in self.entries:
            if nm == name:
                return buf[off:off + n].view(shape)
        raise KeyError(name)

    def init_states(self):
        if not self.trainable:
            return
        lo = self.rank * self.shard
        self.master = self.w[lo:lo + self.shard].to(torch.float32).clone()
        self.m = torch.zeros_like(self.master)
        self.v = torch.zeros_like(self.master)
        self.gshard = torch.empty_like(self.master) if self.world > 1 else None


class DecoderTrainer:
    """One optimizer step = `zero_grad()`, one or more `forward_backward(...)` (gradients accumulate), `step()`.

    model: ufvideo_amd VideoReferQwen2ForCausalLM on a GPU.  After construction the model's packed weights ARE the trainer's
    flat buffers, so generate()/forward() see every update.  group: torch.distributed process group (None = default group if
    initialised, else single process)."""

    lora_bucket = None                  # (class default: set by __init__ when lora= is given)

    def __init__(self, model, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, group=None,
                 train_embed=True, train_projector=False, train_region_encoder=False, train_decoder=True, mm_projector_lr=None,
                 train_seg_head=False, lora=None, optimizer_states=True, gradient_checkpointing=False):
        """gradient_checkpointing=True (the reference's --gradient_checkpointing True, scripts/train/train_1121v1.sh; HF `gradient_checkpointing_enable()` wraps every
        decoder layer): only every layer's INPUT stream is kept from the forward pass; the backward of a layer first re-runs that layer's forward (same kernels, same
        bits) into ONE stash shared by all layers.  Activation memory n_layers x 420 MB -> 34 MB per layer + one 420 MB stash at S = 2399 (7B dimensions), for one more
        forward pass of time; losses and gradients are bit-identical to the stashing form (tests/test_train_gpu.py).
        optimizer_states=False: the forward + backward engine only -- no fp32 masters / Adam moments are allocated and step() raises; the gradients are
        read with export_grad_dict() (this is what `forward(inference=False)` under autograd builds, model/videorefer_qwen2.py: the optimizer is then the
        caller's, e.g. the HF Trainer's, on the model's own nn.Parameters).
        lora = dict(r=8, alpha=16[, seed=0 | init={name: tensor}]): the reference's --lora_enable stage (train.py:829-845: peft LoraConfig over
        find_all_linear_names = the q_proj and v_proj Linears outside the multimodal modules, videorefer_trainer.py:75-90): the language model is frozen
        and y = W x + (alpha / r) B (A x) trains A [r, in] (kaiming-uniform) and B [out, r] (zeros) of every layer's q_proj and v_proj; peft is absent from
        this image, its published LoRA forward is restated (dropout 0: lora_dropout is a training-time random mask that no restatement can reproduce).
        Implies train_decoder=False.  detach() / sync_to_model() merge the adapters into the decoder weights; export_lora_state_dict() returns them
        under peft's names.
        train_decoder=False freezes the language model (the reference's tune_mm_mlp_adapter / tune_region_encoder stages,
        train.py:882-890: model.requires_grad_(False), then only the adapter's parameters are re-enabled): backward still carries
        dL/dx through every layer, but no weight gradient, no fp32 states and no update exist for the decoder."""
        self.model = model
        self.cfg = cfg = model.config
        self.lr, self.betas, self.eps, self.wd, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        # the reference's optimizer groups give mm_projector.* its own learning rate when --mm_projector_lr is set
        # (videorefer_trainer.py:278-306); None = the common rate.  set_lr_ratio() scales both, as its scheduler does per group.
        self.base_lr, self.mm_projector_lr = lr, mm_projector_lr
        self.base_mm_projector_lr = mm_projector_lr
        self.group = group
        self.optimizer_states = bool(optimizer_states)
        self.gradient_checkpointing = bool(gradient_checkpointing)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if not self.optimizer_states:
            # The gradient engine of the autograd path is RANK-LOCAL whatever process group exists: under HF Trainer / torchrun / DDP (the reference's
            # deployment) the averaging of `.grad` is DDP's, through its hooks on the nn.Parameters -- an engine that also issued reduce-scatters inside
            # forward would compete with them (and has no shards to reduce into: init_states() is skipped).  So: world 1, no side stream, no
            # split-K toggle, no _reduce_async / _exchange.
            self.world = 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.t = 0
        self.lora = dict(lora) if lora else None
        self.train_decoder = bool(train_decoder) and not self.lora
        self.train_embed = train_embed = bool(train_embed) and self.train_decoder
        if not self.train_decoder and not (train_projector or train_region_encoder or train_seg_head or self.lora):
            raise ValueError("nothing to train: train_decoder=False needs train_projector, train_region_encoder, train_seg_head and / or lora")
        dev = model.device
        self.dev = dev
        D, I = cfg.hidden_size, cfg.intermediate_size
        H, KV, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        self.QW = (H + 2 * KV) * hd
        pk = model.get_model().packed()
        self.pk = pk
        self.inv_freq = pk["inv_freq"]
        # ---- buckets: one per layer (bf16 matrices), one for the head (lm_head, embed), one fp32 for norms / biases
        self.layers = []
        for L in pk["layers"]:
            if "wqkv8" in L:
                raise NotImplementedError("training runs on the bf16 weights (set_gemm_dtype('bf16'))")
            b = _Bucket([("wqkv", (self.QW, D)), ("wo", (D, H * hd)), ("wgu", (2 * I, D)), ("wd", (D, I))], dev, self.world, self.rank,
                        torch.bfloat16, True, trainable=self.train_decoder)
            for k in ("wqkv", "wo", "wgu", "wd"):
                b.view(b.w, k).copy_(L[k])
                L[k] = b.view(b.w, k)                             # the model now reads the trainer's buffer
            self.layers.append(b)
        V = cfg.vocab_size
        self.V, self.Vp = V, _ru(V, 128)
        hpk = model.packed()
        head = [("lm_head", (self.Vp, D))] + ([("embed", (V, D))] if train_embed else [])
        self.head = _Bucket(head, dev, self.world, self.rank, torch.bfloat16, True, trainable=self.train_decoder)
        self.head.view(self.head.w, "lm_head")[:V].copy_(hpk["lm_head"])
        hpk["lm_head_pad"] = self.head.view(self.head.w, "lm_head")
        hpk["lm_head"] = hpk["lm_head_pad"][:V]
        if train_embed:
            self.head.view(self.head.w, "embed").copy_(pk["embed"])
            pk["embed"] = self.head.view(self.head.w, "embed")
        small = [("norm", (D,))]
        for i in range(len(self.layers)):
            small += [(f"ln1.{i}", (D,)), (f"ln2.{i}", (D,)), (f"bqkv.{i}", (self.QW,))]
        # replicated: all-reduced, updated by every rank (frozen decoder: the gradient buffer is only the kernels' scratch output)
        self.small = _Bucket(small, dev, 1, 0, torch.float32, False, trainable=self.train_decoder, keep_grad=True)
        self.small.view(self.small.w, "norm").copy_(pk["norm"]); pk["norm"] = self.small.view(self.small.w, "norm")
        for i, L in enumerate(pk["layers"]):
            for k in ("ln1", "ln2", "bqkv"):
                self.small.view(self.small.w, f"{k}.{i}").copy_(L[k])
                L[k] = self.small.view(self.small.w, f"{k}.{i}")
        # ---- optional: LoRA adapters on q_proj / v_proj.  Ranks are padded to 128 (rows of A / columns of B beyond r are zero and stay zero: their
        # gradients are products with those zeros) so that every adapter product runs on the MFMA GEMM kernels; replicated fp32 bucket like `small`.
        self.lora_bucket = None
        if self.lora:
            r, alpha = int(self.lora.get("r", 8)), float(self.lora.get("alpha", 16))
            self.lora_r, self.lora_scale, self.Rp = r, alpha / r, _ru(r, 128)
            Rp = self.Rp
            names = []
            for i in range(len(self.layers)):
                names += [(f"Acat.{i}", (2 * Rp, D)), (f"Bq.{i}", (H * hd, Rp)), (f"Bv.{i}", (KV * hd, Rp))]
            self.lora_bucket = lb = _Bucket(names, dev, 1, 0, torch.float32, True)
            init = self.lora.get("init")
            gen = torch.Generator().manual_seed(int(self.lora.get("seed", 0)))
            for i in range(len(self.layers)):
                a = lb.view(lb.w, f"Acat.{i}")
                for j, nm in enumerate(("q_proj", "v_proj")):
                    key = f"model.layers.{i}.self_attn.{nm}.lora_A.weight"
                    if init is not None and key in init:
                        a[j * Rp:j * Rp + r].copy_(init[key])
                    else:                                           # peft: kaiming_uniform_(a = sqrt(5)) = U(-1/sqrt(in), 1/sqrt(in))
                        a[j * Rp:j * Rp + r].copy_((torch.rand(r, D, generator=gen) * 2 - 1) / math.sqrt(D))
                    keyb = f"model.layers.{i}.self_attn.{nm}.lora_B.weight"
                    if init is not None and keyb in init:
                        lb.view(lb.w, f"B{nm[0]}.{i}")[:, :r].copy_(init[keyb])
            self.lora_wb = torch.zeros((lb.n,), device=dev, dtype=torch.bfloat16)           # bf16 working copies + the transposes the backward reads
            self.lora_T = [dict(AcatT=torch.empty((D, 2 * Rp), device=dev, dtype=torch.bfloat16), BqT=torch.empty((Rp, H * hd), device=dev, dtype=torch.bfloat16),
                                BvT=torch.empty((Rp, KV * hd), device=dev, dtype=torch.bfloat16)) for _ in self.layers]
        # ---- optional: the multimodal projector (reference: mm_projector is in the trainable set, train.py:873-912).  Its
        # parameters keep the reference layout (the connector re-packs them after every step); replicated update like `small`.
        self.pgrad = None
        self.proj_bucket = None                # "auxiliary" bucket: projector and / or region encoder, replicated update
        self.aux_modules = {}
        if train_projector:
            from .train_projector import ProjectorGrad
            self.pgrad = ProjectorGrad(model.get_model().mm_projector)
            self.aux_modules["mm_projector."] = model.get_model().mm_projector
        self.train_region = bool(train_region_encoder)
        if self.train_region:
            self.aux_modules["region_encoder."] = model.get_model().region_encoder
        # the mask branch (reference: text_hidden_fcs always trainable, sam_mask_decoder with train_mask_decoder, videorefer_arch.py:124-149):
        # trained through BCE + DICE of the [SEG] masks (ufvideo_amd/train_seg.py); same replicated fp32 bucket as the other adapters
        self.seg = None
        if train_seg_head:
            from .train_seg import SegHeadGrad
            inner = model.get_model()
            if inner.mask_encoder is None:
                raise ValueError("train_seg_head needs the SAM2 head: build the model with config.sam2_trunk set")
            self.seg = SegHeadGrad(model)
            self.aux_modules["text_hidden_fcs."] = inner.text_hidden_fcs
            self.aux_modules["mask_encoder.sam2_model.sam_mask_decoder."] = inner.mask_encoder.sam2_model.sam_mask_decoder
        if self.aux_modules:
            named = [(pre + k, v) for pre, mod in self.aux_modules.items() for k, v in mod.named_parameters()]
            # per module: matrices first, then vectors, so that every (learning rate, decay) group is ONE contiguous range of the flat
            # buffers and the update is a handful of launches instead of one per parameter (the connector has 113)
            named.sort(key=lambda kv: (not kv[0].startswith("mm_projector."), kv[1].ndim < 2))
            self.proj_params = dict(named)
            self.proj_bucket = pb = _Bucket([(k, tuple(v.shape)) for k, v in named], dev, 1, 0, torch.float32, True)
            self.proj_nodecay = [k for k, v in named if v.ndim < 2]
            self.proj_ranges = []                                  # [lo, hi, is_projector, decay]; alignment gaps hold zeros and stay zero
            for name, shape, off, n in pb.entries:
                key = (name.startswith("mm_projector."), name not in self.proj_nodecay)
                if self.proj_ranges and self.proj_ranges[-1][2:] == list(key):
                    self.proj_ranges[-1][1] = off + n
                else:
                    self.proj_ranges.append([off, off + n, *key])
            # the modules' (bf16) parameters become views of one flat buffer: refreshed from the masters with one conversion per step
            same = len({v.dtype for _, v in named}) == 1
            self.proj_flat = torch.zeros((pb.n,), device=dev, dtype=named[0][1].dtype) if same else None
            for (k, v), (_, shape, off, n) in zip(named, pb.entries):
                pb.view(pb.w, k).copy_(v)
                if same:
                    self.proj_flat[off:off + n].view(shape).copy_(v)
                    v.data = self.proj_flat[off:off + n].view(shape)
        if self.optimizer_states:
            for b in self.buckets():
                b.init_states()
        if self.lora_bucket is not None:
            self._refresh_lora()
        # ---- transposed weight copies for dX = dY W (the NT GEMM wants W^T rows)
        self.wT = [dict(wqkv=torch.empty((D, self.QW), device=dev, dtype=torch.bfloat16),
                        wo=torch.empty((H * hd, D), device=dev, dtype=torch.bfloat16),
                        wgu=torch.empty((D, 2 * I), device=dev, dtype=torch.bfloat16),
                        wd=torch.empty((I, D), device=dev, dtype=torch.bfloat16)) for _ in self.layers]
        self.lm_headT = torch.empty((D, self.Vp), device=dev, dtype=torch.bfloat16)
        self._refresh_transposes()
        self._stash_S = 0
        # flash-style attention backward (csrc/attn_bwd.hip) for head_dim 128; UFV_TRAIN_ATTN=materialised keeps the per-group GEMM pipeline
        self.fused_attn_bwd = hd == 128 and os.environ.get("UFV_TRAIN_ATTN", "fused") != "materialised"
        self._fresh = True
        self._last_micro = False
        self.comm_stream = torch.cuda.Stream(device=dev) if self.world > 1 else None
        if self.world > 1:
            # RCCL kernels on the side stream hold CUs while GEMMs run: the split-K GEMM form spins on turn flags and assumes that its <= 256 blocks all
            # become resident -- keep ufv_gemm on the unsplit kernels in a process that overlaps collectives with compute (threading)
            # (ufv_gemm_set_splitk: an atomic in the library, restored by detach().  The unsplit form sums K in one pass instead of in parts, so
            # fp32 GEMM outputs of this process differ from a world-1 process in the last bits, <= 4e-6 of the largest element.)
            from . import _lib as _L
            self._splitk_before = _L.load().ufv_gemm_set_splitk(0)
        # the decoder's packed weights now ARE this trainer's buffers: anything that would re-pack them from the (stale) nn.Parameters
        # -- .to(), load_state_dict, set_gemm_dtype, resize_token_embeddings, invalidate() -- raises until detach()
        model.get_model()._owner = self
        model._owner = self

    def buckets(self):
        """the buckets that are exchanged and updated"""
        dec = self.layers + [self.head, self.small] if self.train_decoder else []
        return dec + ([self.lora_bucket] if self.lora_bucket is not None else []) + ([self.proj_bucket] if self.proj_bucket is not None else [])

    def _lora_w(self, i):
        lb = self.lora_bucket
        return lb.view(self.lora_wb, f"Acat.{i}"), lb.view(self.lora_wb, f"Bq.{i}"), lb.view(self.lora_wb, f"Bv.{i}")

    def _refresh_lora(self):
        """bf16 working copies of the adapters from the fp32 values, and the transposes the backward's NT GEMMs read"""
        self.lora_wb.copy_(self.lora_bucket.w)
        for i, t in enumerate(self.lora_T):
            a, bq, bv = self._lora_w(i)
            ops.transpose(a, rpad=a.shape[0], out=t["AcatT"]); ops.transpose(bq, rpad=bq.shape[0], out=t["BqT"]); ops.transpose(bv, rpad=bv.shape[0], out=t["BvT"])

    def _refresh_transposes(self):
        for b, t in zip(self.layers, self.wT):
            for k in ("wqkv", "wo", "wgu", "wd"):
                w = b.view(b.w, k)
                ops.transpose(w, rpad=w.shape[0], out=t[k])
        ops.transpose(self.head.view(self.head.w, "lm_head"), rpad=self.Vp, out=self.lm_headT)

    # ---- activation stash -------------------------------------------------------------------------------------
    def _alloc(self, S):
        if S <= self._stash_S:
            return
        cfg, dev = self.cfg, self.dev
        D, I = cfg.hidden_size, cfg.intermediate_size
        H, KV, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        Sp = _ru(S, 128)
        bf, f32 = torch.bfloat16, torch.float32
        def stash():
            st = dict(h1=torch.empty((S, D), device=dev, dtype=bf),
                      qkv=torch.empty((S, self.QW), device=dev, dtype=bf), kv=torch.zeros((Sp, 2 * KV * hd), device=dev, dtype=bf),
                      o=torch.empty((S, H * hd), device=dev, dtype=bf), x_mid=torch.empty((S, D), device=dev, dtype=f32),
                      h2=torch.empty((S, D), device=dev, dtype=bf), gu=torch.empty((S, 2 * I), device=dev, dtype=bf),
                      act=torch.empty((S, I), device=dev, dtype=bf), lse=torch.empty((H, S), device=dev, dtype=f32))
            if self.lora_bucket is not None:
                st["u"] = torch.empty((S, 2 * self.Rp), device=dev, dtype=bf)              # A x of both adapters, kept for dB
            return st
        self.st = None                                                                      # release the old stashes before the new ones are allocated
        if self.gradient_checkpointing:                  # ONE stash, refilled by the layer's re-run forward in front of its backward; per layer only the input stream
            shared = stash()
            self.st = [dict(shared, x_in=torch.empty((S, D), device=dev, dtype=f32)) for _ in self.layers]
        else:
            self.st = [dict(stash(), x_in=torch.empty((S, D), device=dev, dtype=f32)) for _ in self.layers]
        # scratch shared by all layers
        W = max(2 * I, self.QW, D, H * hd, 2 * self.Rp if self.lora_bucket is not None else 0)      # (the LoRA backward transposes [S, 2 Rp] into inT / dyT)
        self.sc = dict(dxb=torch.empty((S, D), device=dev, dtype=bf), dxbT=torch.empty((D, Sp), device=dev, dtype=bf),
                       inT=torch.empty((W, Sp), device=dev, dtype=bf), dyT=torch.empty((W, Sp), device=dev, dtype=bf),
                       dact=torch.empty((S, I), device=dev, dtype=bf), dgu=torch.empty((S, 2 * I), device=dev, dtype=bf),
                       dh=torch.empty((S, D), device=dev, dtype=f32), do=torch.empty((S, H * hd), device=dev, dtype=bf),
                       dqkv=torch.empty((S, self.QW), device=dev, dtype=bf))
        if self.lora_bucket is not None:
            self.sc["du"] = torch.empty((S, 2 * self.Rp), device=dev, dtype=bf)
        self._stash_S = S

    def zero_grad(self):
        """Start a new accumulation window.  The matrix gradients are not cleared: the first forward_backward() of the window
        OVERWRITES them (its dW GEMMs run without the fp32 residual input), which saves one write and one read of the 30 GB
        gradient buffers per step; only the small fp32 gradients and the embedding rows (scatter-add) are zeroed."""
        self.small.g.zero_()
        if self.proj_bucket is not None:
            self.proj_bucket.g.zero_()
        if self.train_embed:
            self.head.view(self.head.g, "embed").zero_()
        self._fresh = True

    # ---- forward with stash + backward ---------------------------------------------------------------------------
    def forward_backward(self, inputs_embeds, labels, embed_ids=None, loss_weight=None, last=False, hidden_hook=None):
        """inputs_embeds fp32 [S, D] (one spliced sample), labels int64 [S] ALREADY SHIFTED (labels[p] = target of position p,
        -100 = ignored).  embed_ids int64 [S]: vocabulary row of every position that came from embed_tokens, -1 elsewhere
        (visual / region tokens).  loss_weight: d(total loss)/d(sum of token losses), default 1 / (valid labels of this sample).
        last=True: no further micro-batch follows before step(), so every layer's gradient bucket is handed to the exchange
        as soon as that layer's backward is done (overlapped reduce-scatter).
        hidden_hook(hb) -> (rows int64 [n], grads fp32 [n, D]) | None: called between forward and backward with the final-norm hidden
        states hb bf16 [S, D]; what it returns is added to d(loss)/d(hb) at those rows (the mask losses reach the decoder through
        the hidden states in front of the [SEG] targets).
        Accumulates gradients; returns (loss = loss_weight * sum CE, d_inputs_embeds fp32 [S, D])."""
        self._last_micro = bool(last)
        cfg = self.cfg
        S, D = inputs_embeds.shape
        I = cfg.intermediate_size
        H, KV, hd, eps = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.rms_norm_eps
        self._alloc(S)
        Sp = _ru(S, 128)
        pk = self.pk
        x = inputs_embeds.to(torch.float32).contiguous().clone()
        rope_tab = ops.rope_table(self.inv_freq, 0, S, hd) if hd % 16 == 0 else None
        # ---------------- forward (same kernels as inference; gate/up kept un-fused so the pre-activations are stashed)
        def layer_forward(lf, x):
            """one decoder layer on the stream x (fp32 [S, D], updated in place), every intermediate the backward reads written into the layer's stash"""
            L, st = pk["layers"][lf], self.st[lf]
            h1, qkv, kv, o, h2, gu, act = st["h1"][:S], st["qkv"][:S], st["kv"], st["o"][:S], st["h2"][:S], st["gu"][:S], st["act"][:S]
            ops.rmsnorm(x, L["ln1"], eps, out=h1)
            ops.gemm(h1, L["wqkv"], bias=L["bqkv"], out=qkv)
            if self.lora_bucket is not None:                                          # q += s B_q (A_q h), v += s B_v (A_v h); every product rounded to bf16
                a_cat, b_q, b_v = self._lora_w(lf)
                u, Rp = st["u"][:S], self.Rp
                ops.gemm(h1, a_cat, out=u)
                qkv[:, :H * hd].add_(ops.gemm(u[:, :Rp], b_q), alpha=self.lora_scale)
                qkv[:, (H + KV) * hd:].add_(ops.gemm(u[:, Rp:], b_v), alpha=self.lora_scale)
            ops.rope_kv(qkv, S, H, KV, hd, self.inv_freq, 0, kv, table=rope_tab)
            if self.fused_attn_bwd:                                                  # same kernel + the log-sum-exp the backward needs
                ops.attention_causal_lse(qkv, kv, kv[:, KV * hd:], o, st["lse"], S, H, KV, hd)
            else:
                ops.attention(qkv, kv, kv[:, KV * hd:], 1, H, KV, S, S, hd, (0, qkv.stride(0)), (0, kv.stride(0)), (0, kv.stride(0)),
                              causal=True, q_pos0=0, out=o)
            ops.gemm(o, L["wo"], resid=x, out=x)
            st["x_mid"][:S].copy_(x)
            ops.rmsnorm(x, L["ln2"], eps, out=h2)
            ops.gemm(h2, L["wgu"], out=gu)
            ops.swiglu(gu, out=act)
            ops.gemm(act, L["wd"], resid=x, out=x)

        for lf in range(len(self.layers)):
            self.st[lf]["x_in"][:S].copy_(x)
            layer_forward(lf, x)
        x_last = x
        hb = ops.rmsnorm(x_last, pk["norm"], eps)                                   # bf16 [S, D]
        lm = self.head.view(self.head.w, "lm_head")
        logits = ops.gemm(hb, lm, out_dtype=torch.float32)                           # [S, Vp]
        labels = labels.to(self.dev).contiguous()
        n_valid = int((labels != -100).sum().item())
        gscale = (1.0 / max(n_valid, 1)) if loss_weight is None else float(loss_weight)
        dl = torch.empty((S, self.Vp), device=self.dev, dtype=torch.bfloat16)
        loss_rows, dl = ops.cross_entropy_bwd(logits, labels, self.V, gscale, dlogits=dl)
        loss = loss_rows.sum() * gscale
        del logits
        # ---------------- backward
        sc = self.sc
        fresh = self._fresh
        dxb, dxbT, inT, dyT, dh = sc["dxb"][:S], sc["dxbT"], sc["inT"], sc["dyT"], sc["dh"][:S]
        g_small = self.small.g

        td = self.train_decoder

        def dW(b_or_buf, name, dy_T, x_T, rows, cols):
            """grad[name] [rows, cols] += dy^T [rows, Sp] . (x^T [cols, Sp])^T"""
            gw = b_or_buf.view(b_or_buf.g, name)
            ops.gemm(dy_T[:rows], x_T[:cols], resid=None if fresh else gw, out=gw)

        # lm_head + final norm
        dx = torch.empty((S, D), device=self.dev, dtype=torch.float32)
        ops.gemm(dl, self.lm_headT, out=dh)                                          # d hb, fp32 [S, D]
        if hidden_hook is not None:
            extra = hidden_hook(hb)
            if extra is not None:
                dh.index_add_(0, extra[0].to(self.dev), extra[1].to(torch.float32))
        if td:
            dlT = ops.transpose(dl, rpad=Sp)                                         # [Vp, Sp]
            hbT = ops.transpose(hb, rpad=Sp, out=inT)
            dW(self.head, "lm_head", dlT, hbT, self.Vp, D)
            del dlT
        del dl
        ops.rmsnorm_bwd(x_last, pk["norm"], dh, dx, self.small.view(g_small, "norm"), eps, accumulate=False)
        xr = torch.empty((S, D), device=self.dev, dtype=torch.float32) if self.gradient_checkpointing else None
        for li in range(len(self.layers) - 1, -1, -1):
            L, st, b, wT = pk["layers"][li], self.st[li], self.layers[li], self.wT[li]
            if self.gradient_checkpointing and li < len(self.layers) - 1:           # (the last layer's intermediates are still in the shared stash from the forward pass)
                xr.copy_(st["x_in"][:S])
                layer_forward(li, xr)                                                # the same launches on the same input: the same bits
            h1, qkv, kv, o, h2, gu, act = st["h1"][:S], st["qkv"][:S], st["kv"], st["o"][:S], st["h2"][:S], st["gu"][:S], st["act"][:S]
            # ---- MLP: x_out = x_mid + down(act)
            ops.convert_into(dx, dxb)
            ops.gemm(dxb, wT["wd"], out=sc["dact"][:S])                              # dact [S, I]
            if td:
                ops.transpose(dxb, rpad=Sp, out=dxbT)
                ops.transpose(act, rpad=Sp, out=inT)
                dW(b, "wd", dxbT, inT, D, I)
            ops.swiglu_bwd(gu, sc["dact"][:S], out=sc["dgu"][:S])
            ops.gemm(sc["dgu"][:S], wT["wgu"], out=dh)                               # d h2, fp32
            if td:
                ops.transpose(sc["dgu"][:S], rpad=Sp, out=dyT)
                ops.transpose(h2, rpad=Sp, out=inT)
                dW(b, "wgu", dyT, inT, 2 * I, D)
            ops.rmsnorm_bwd(st["x_mid"][:S], L["ln2"], dh, dx, self.small.view(g_small, f"ln2.{li}"), eps)
            # ---- attention: x_mid = x_in + o_proj(attn)
            ops.convert_into(dx, dxb)
            ops.gemm(dxb, wT["wo"], out=sc["do"][:S])                                # d o [S, H*hd]
            if td:
                ops.transpose(dxb, rpad=Sp, out=dxbT)
                ops.transpose(o, rpad=Sp, out=inT)
                dW(b, "wo", dxbT, inT, D, H * hd)
            dqkv = sc["dqkv"][:S]
            if self.fused_attn_bwd:
                ops.attention_bwd_fused(qkv, kv, kv[:, KV * hd:], o, sc["do"][:S], st["lse"], dqkv, dqkv[:, H * hd:],
                                        dqkv[:, (H + KV) * hd:], S, H, KV, hd)
            else:
                ops.attention_bwd(qkv, kv, kv[:, KV * hd:], sc["do"][:S], dqkv, dqkv[:, H * hd:], dqkv[:, (H + KV) * hd:], S, H, KV, hd)
            ops.rope_rows(dqkv, 0, H + KV, hd, self.inv_freq, 0, backward=True)
            ops.gemm(dqkv, wT["wqkv"], out=dh)                                       # d h1, fp32
            if self.lora_bucket is not None:
                lb, lt, Rp, u, du = self.lora_bucket, self.lora_T[li], self.Rp, st["u"][:S], sc["du"][:S]
                a_cat, b_q, b_v = self._lora_w(li)
                ddq = dqkv[:, :H * hd] * self.lora_scale                             # d(s B u) / d(B u), bf16
                ddv = dqkv[:, (H + KV) * hd:] * self.lora_scale
                ops.gemm(ddq, lt["BqT"], out=du[:, :Rp])                             # d u = dd B
                ops.gemm(ddv, lt["BvT"], out=du[:, Rp:])
                uT = ops.transpose(u, rpad=Sp, out=inT)                              # [2 Rp, Sp]
                dW(lb, f"Bq.{li}", ops.transpose(ddq, rpad=Sp, out=dyT), uT[:Rp], H * hd, Rp)
                dW(lb, f"Bv.{li}", ops.transpose(ddv, rpad=Sp, out=dyT), uT[Rp:2 * Rp], KV * hd, Rp)
                duT = ops.transpose(du, rpad=Sp, out=dyT)
                dW(lb, f"Acat.{li}", duT, ops.transpose(h1, rpad=Sp, out=inT), 2 * Rp, D)
                ops.gemm(du, lt["AcatT"], resid=dh, out=dh)                          # d h1 += d u A
            if td:
                ops.transpose(dqkv, rpad=Sp, out=dyT)
                ops.transpose(h1, rpad=Sp, out=inT)
                dW(b, "wqkv", dyT, inT, self.QW, D)
                ops.colsum(dqkv, self.small.view(g_small, f"bqkv.{li}"))
            ops.rmsnorm_bwd(st["x_in"][:S], L["ln1"], dh, dx, self.small.view(g_small, f"ln1.{li}"), eps)
            if td:
                self._reduce_async(b)
        self._fresh = False
        if self.train_embed and embed_ids is not None:
            ops.scatter_add_rows(dx, embed_ids.to(self.dev).contiguous(), self.head.view(self.head.g, "embed"))
        return loss, dx

    def loss_and_grads(self, input_ids=None, labels=None, attention_mask=None, images=None, masks=None, frame=None, ann_indices=None,
                       frame_nums=None, video_file=None, **_unused):
        """Forward + backward of one collated batch (the keys the reference's collator produces, train.py:706-732; SURVEY
        §3.5) into the gradient buffers, no update: splice (the tower is frozen; projector / region encoder are trained when the trainer was built with
        train_projector / train_region_encoder, else they run forward-only),
        causal-LM loss averaged over the batch's supervised tokens (HF Qwen2ForCausalLM), backward, exchange, AdamW.
        The remaining collator keys (images_sam, offset, masks_list, label_list) are accepted; a batch whose labels contain [SEG] or
        whose masks_list holds ground-truth masks raises NotImplementedError (see below).  Returns {"loss", "ce_loss", ...}: for
        the batches that are accepted the mask terms of the reference's loss are zero, so "loss" is the reference's loss."""
        m = self.model
        # The reference's objective adds bce_loss_weight * BCE + dice_loss_weight * DICE of the SAM2 masks for every [SEG] in the labels
        # (videorefer_qwen2.py:198-352) and trains text_hidden_fcs / the mask decoder through it: built in ufvideo_amd/train_seg.py and
        # enabled with train_seg_head=True.  A trainer without it refuses such a batch instead of training it on the CE term alone
        # under the name "loss".
        masks_list = _unused.get("masks_list")
        images_sam, label_list, offset = _unused.get("images_sam"), _unused.get("label_list"), _unused.get("offset")
        seg_id = getattr(self.cfg, "seg_token_id", None)
        has_gt = masks_list is not None and any(torch.is_tensor(g) and g.numel() > 0 and g.shape[0] > 0 for g in masks_list)
        has_seg = seg_id is not None and labels is not None and bool((labels == seg_id).any())
        if (has_gt or has_seg) and self.seg is None:
            raise NotImplementedError("train_step: this batch carries [SEG] targets / ground-truth masks; build the trainer with train_seg_head=True "
                                      "to train the mask-loss (BCE + DICE) branch -- without it only the causal-LM objective is trained")
        stashes = None
        with torch.no_grad():
            mm_features = None
            if self.pgrad is not None:                 # projector forward with every pre-activation kept, one stash per video
                nf = m.num_frames()
                vids = torch.stack([d.expand(nf, -1, -1, -1) if modal == "image" else d for d, modal in images], 0)
                Bv, T = vids.shape[:2]
                feats = m.get_model().get_vision_tower().encode(vids.reshape(Bv * T, *vids.shape[2:]))      # frozen tower
                feats = feats.view(Bv, T, feats.shape[1], feats.shape[2])
                hw = int(feats.shape[2] ** 0.5)
                outs, stashes = [], []
                for i in range(Bv):
                    o, st_ = self.pgrad.forward(feats[i].reshape(T * hw * hw, feats.shape[3]), T, hw)
                    outs.append(o); stashes.append(st_)
                mm_features = torch.stack(outs, 0)
            region_stash = {} if (self.train_region and frame is not None) else None
            (_, am, _, embeds, new_labels, _) = m.prepare_inputs_labels_for_multimodal(input_ids, attention_mask, None, labels, images, masks,
                                                                                      frame, ann_indices, frame_nums, video_file,
                                                                                      mm_features=mm_features, region_stash=region_stash)
        if embeds is None:
            raise ValueError("train_step needs multimodal inputs (images=...) as the reference's training batches have")
        eids = m._last_embed_ids
        B, S, _ = embeds.shape
        shifted, lens = [], []
        for b in range(B):
            n = int(am[b].sum().item()) if am is not None else S
            lab = new_labels[b, :n].to("cpu")
            shifted.append(torch.cat([lab[1:], torch.full((1,), -100, dtype=lab.dtype)]))
            lens.append(n)
        n_valid = sum(int((s_ != -100).sum()) for s_ in shifted)
        ce_w = float(getattr(self.cfg, "ce_loss_weight", 1.0))
        w = ce_w / max(n_valid, 1)
        self.zero_grad()
        loss = torch.zeros((), device=self.dev)
        mask_bce = torch.zeros((), device=self.dev)
        mask_dice = torch.zeros((), device=self.dev)
        need_dx = stashes is not None or region_stash is not None
        dxs = torch.zeros((B * S, embeds.shape[2]), device=self.dev, dtype=torch.float32) if need_dx else None
        seg_leaves = None
        if self.seg is not None and (has_gt or has_seg):
            # fp32 leaves over the masters of the mask branch; their .grad is folded into the auxiliary bucket after the last sample
            # (the gradient engine of the autograd path has no masters: optimizer_states=False -- its leaves are fp32 copies of the working weights)
            pb = self.proj_bucket
            src_buf = pb.master if pb.master is not None else pb.w
            seg_leaves = {k: pb.view(src_buf, k).detach().float().requires_grad_(True) for k in self.proj_params
                          if k.startswith(("text_hidden_fcs.", "mask_encoder."))}
            if torch.is_tensor(offset):
                offset = offset.tolist()
            offset = list(range(B + 1)) if offset is None else [int(o) for o in offset]
            assert len(offset) == B + 1 and offset == list(range(B + 1)), "train_step: one conversation per sample (offset = [0, 1, ..., B])"
            num_masks_total = sum(int(g.shape[0]) for g in masks_list)
            w_bce, w_dice = float(getattr(self.cfg, "bce_loss_weight", 1.0)), float(getattr(self.cfg, "dice_loss_weight", 1.0))
        for b in range(B):
            hook = None
            if seg_leaves is not None:
                rows = torch.nonzero(shifted[b] == seg_id).reshape(-1)          # position p is queried when label p + 1 is [SEG]
                if rows.numel():
                    def hook(hb, b=b, rows=rows):
                        nonlocal mask_bce, mask_dice
                        # the mask branch differentiates through torch autograd (train_seg.py: _Cast / lin nodes over the HIP kernels); this hook also runs
                        # inside _TrainingLoss.forward, where grad mode is off: switch it on locally or no graph is recorded and hid.grad stays None
                        with torch.enable_grad():
                            hid = hb[rows.to(hb.device)].detach().float().requires_grad_(True)
                            hw = tuple(label_list[b].shape)
                            bce_b, dice_b = self.seg.forward_backward(seg_leaves, hid, images_sam[b], masks_list[b], hw, w_bce, w_dice, num_masks_total)
                        if hid.grad is None:
                            raise RuntimeError("mask-loss branch: no gradient reached the [SEG] hidden states")
                        mask_bce, mask_dice = mask_bce + bce_b.detach(), mask_dice + dice_b.detach()
                        return rows, hid.grad
                else:
                    assert masks_list[b].shape[0] == 0, f"gt_mask.shape: {tuple(masks_list[b].shape)}, pred_mask.shape: (0, ...)"
            l_b, dx_b = self.forward_backward(embeds[b, :lens[b]], shifted[b], embed_ids=eids[b, :lens[b]], loss_weight=w, last=(b == B - 1),
                                              hidden_hook=hook)
            loss = loss + l_b
            if dxs is not None:
                dxs[b * S:b * S + lens[b]].copy_(dx_b)
        if stashes is not None:                        # visual-token rows of d(inputs_embeds) -> projector backward, per video
            m_src, m_dst, n_mm, tok = m._last_mm_map
            d_mm = torch.zeros((n_mm * tok, embeds.shape[2]), device=self.dev, dtype=torch.float32)
            i64 = lambda l: torch.tensor(l, dtype=torch.int64, device=self.dev)
            if m_src:
                ops.gather_rows(dxs, i64(m_dst), d_mm, i64(m_src))
            for k, st_ in enumerate(stashes):
                grads, _ = self.pgrad.backward(d_mm[k * tok:(k + 1) * tok], st_)
                names = ["mm_projector." + name for name in grads]
                torch._foreach_add_([self.proj_bucket.view(self.proj_bucket.g, nm) for nm in names],
                                    [gval.reshape(self.proj_params[nm].shape) for nm, gval in zip(names, grads.values())])
        if region_stash:                               # region-token rows -> the region encoder's MLP
            r_src, r_dst, n_reg = m._last_region_map
            if r_src:
                i64 = lambda l: torch.tensor(l, dtype=torch.int64, device=self.dev)
                d_reg = torch.zeros((n_reg, embeds.shape[2]), device=self.dev, dtype=torch.float32)
                ops.gather_rows(dxs, i64(r_dst), d_reg, i64(r_src))
                for name, gval in m.get_model().region_encoder.backward(d_reg, region_stash).items():
                    name = "region_encoder." + name
                    self.proj_bucket.view(self.proj_bucket.g, name).add_(gval.reshape(self.proj_params[name].shape))
        if seg_leaves is not None:
            for k, leaf in seg_leaves.items():
                if leaf.grad is not None:
                    self.proj_bucket.view(self.proj_bucket.g, k).add_(leaf.grad)
        ce = loss                                          # ce_loss_weight is part of the per-token weight (and so of the gradient)
        mask_loss = mask_bce + mask_dice
        return {"loss": ce + mask_loss, "ce_loss": ce, "mask_bce_loss": mask_bce, "mask_dice_loss": mask_dice, "mask_loss": mask_loss}

    def train_step(self, **batch):
        """One optimizer step on a collated batch with the keys the reference's collator produces (train.py:706-732):
        loss_and_grads(**batch), then exchange + clip + AdamW (step()).  Returns the loss terms + "grad_norm"."""
        out = self.loss_and_grads(**batch)
        self.step()
        out["grad_norm"] = getattr(self, "last_grad_norm", None)
        return out

    # ---- data-parallel exchange + update (ZeRO-2) -----------------------------------------------------------------------
    def _reduce_async(self, b):
        """Bucket b's gradients are final once its layer's backward has run in the LAST micro-batch of the window: start its
        reduce-scatter on the side stream so that it overlaps the backward of the earlier layers."""
        if self.world == 1 or not self._last_micro:
            return
        if self.comm_stream is None:                      # no side stream (CPU tensors in the gloo tests): same collective, in line
            reduce_scatter_mean(b.gshard, b.g, self.group)
        else:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                reduce_scatter_mean(b.gshard, b.g, self.group)
        b.reduced = True

    def _exchange_order(self):
        """(sharded buckets in the order backward produced them, replicated buckets): every rank issues its collectives in this
        order -- the property the gloo test pins, since mismatched orders deadlock or mix buffers on any backend"""
        sharded = (list(reversed(self.layers)) + [self.head]) if self.train_decoder else []
        replicated = ([self.small] if self.train_decoder else []) + ([self.lora_bucket] if self.lora_bucket is not None else []) + \
                     ([self.proj_bucket] if self.proj_bucket is not None else [])
        return sharded, replicated

    def _exchange(self):
        if self.world == 1:
            return
        cs = self.comm_stream
        sharded, replicated = self._exchange_order()

        def run():
            for b in sharded:
                if not getattr(b, "reduced", False):              # not already started by _reduce_async
                    reduce_scatter_mean(b.gshard, b.g, self.group)
                b.reduced = False
            for rb in replicated:
                dist.all_reduce(rb.g, op=dist.ReduceOp.SUM, group=self.group)
                rb.g.mul_(1.0 / self.world)
        if cs is None:
            run()
            return
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            run()
        torch.cuda.current_stream().wait_stream(cs)

    def _grad_shards(self):
        out = []
        for b in self.buckets():
            out.append((b, b.g if (self.world == 1 or b.world == 1) else b.gshard))
        return out

    def _check_gemm_errors(self):
        """The per-step guarantee of include/ufv.h (Conventions): the kernels of this step's forward / backward set the device's pinned error word when a
        split-K turn wait expires, so the word means something only AFTER they have run -- the stream is synchronised before it is read (once per optimizer
        step: the update cannot overlap the backward pass anyway, it consumes its gradients), so a bad step is refused before AdamW touches the fp32
        masters, not one step later.  With world > 1 the flag is all-reduced (MAX) BEFORE any rank raises: a rank-local raise in front of _exchange()
        would leave the other ranks waiting in reduce-scatter.  A negative return is the library failing to find the device, reported as that."""
        if self.dev.type == "cuda":
            torch.cuda.current_stream(self.dev).synchronize()
        err = int(_lib.load().ufv_gemm_error_state())
        if self.world > 1:
            flag = torch.tensor([err if err > 0 else (1 << 20 if err < 0 else 0)], device=self.dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            worst = int(flag.item())
            if worst != 0 and err == 0:
                raise _lib.UfvError(f"another rank reported a GEMM error (code {worst}) in this step: every rank refuses the update together")
        if err < 0:
            raise _lib.UfvError(f"ufv_gemm_error_state failed (code {err}: no current HIP device / runtime error) -- this is NOT a split-K timeout: "
                                f"{_lib.load().ufv_last_error().decode()}")
        if err != 0:
            raise _lib.UfvError(f"a split-K GEMM of this step timed out waiting for its turn (error word {err}): the gradients are not trustworthy and "
                                f"the update was NOT applied; ufv_gemm_clear_error() resets the word")

    def step(self):
        """Gradient exchange, global-norm clipping (HF Trainer max_grad_norm), AdamW on this rank's shard, all-gather.  Raises when a split-K GEMM of the
        forward / backward pass left the device's error word set (a timed-out turn wait: that launch's tile is wrong, include/ufv.h Conventions)."""
        if not self.optimizer_states:
            raise RuntimeError("this DecoderTrainer was built with optimizer_states=False (gradient engine of the autograd path): the optimizer is the caller's")
        self._check_gemm_errors()
        self.t += 1
        self._exchange()
        shards = self._grad_shards()
        gscale = None
        if self.max_grad_norm and self.max_grad_norm > 0:
            tot = torch.zeros((), device=self.dev, dtype=torch.float32)
            small_sq = None
            small_sq = torch.zeros((), device=self.dev, dtype=torch.float32)
            for b, g in shards:
                s = ops.sumsq(g).sum()
                if b.world == 1 and self.world > 1:
                    small_sq = small_sq + s
                else:
                    tot = tot + s
            if self.world > 1:
                dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)         # shards are disjoint
            tot = tot + small_sq                                                     # replicated buckets: counted once
            norm = tot.sqrt()
            gscale = torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0).reshape(1).contiguous()
            self.last_grad_norm = norm
        b1, b2 = self.betas
        for b, g in shards:
            wd = self.wd if b.decay else 0.0
            if b is self.proj_bucket:                 # decay on matrices only: two passes over the flat buffer by entry
                for lo, hi, is_proj, decay in self.proj_ranges:
                    lr_e = self.mm_projector_lr if (self.mm_projector_lr is not None and is_proj) else self.lr
                    ops.adamw(b.master[lo:hi], g[lo:hi], b.m[lo:hi], b.v[lo:hi], None, lr_e, b1, b2, self.eps, self.wd if decay else 0.0,
                              self.t, gscale)
                b.w.copy_(b.master)
                if self.proj_flat is not None:
                    self.proj_flat.copy_(b.master)
                else:
                    for k, v in self.proj_params.items():
                        v.data.copy_(b.view(b.w, k))
                for mod in self.aux_modules.values():
                    mod.invalidate()
            elif b is self.small or b is self.lora_bucket:
                ops.adamw(b.master, g, b.m, b.v, None, self.lr, b1, b2, self.eps, wd, self.t, gscale)
                b.w.copy_(b.master)
                if b is self.lora_bucket:
                    self._refresh_lora()
            else:
                lo = b.rank * b.shard
                ops.adamw(b.master, g, b.m, b.v, b.w[lo:lo + b.shard], self.lr, b1, b2, self.eps, wd, self.t, gscale)
        if self.train_decoder:
            if self.world > 1:
                for b in self.layers + [self.head]:
                    all_gather_shards(b.w, self.group)
            self._refresh_transposes()

    # ---- hand the trained weights back / save and resume ----------------------------------------------------------------------
    def sync_to_model(self):
        """Writes the trained decoder weights (bf16 working copies of the fp32 masters) into the model's nn.Parameters under the
        reference's names, so that model.state_dict() / save paths export what was trained.  The projector / region-encoder parameters
        are views of the trainer's buffer already.  Every rank holds the full bf16 weights after step(), so no communication.
        With LoRA adapters the parameters receive the MERGED view W + (alpha / r) B A (what peft's merge_and_unload would export) and
        NOTHING of the trainer changes: the packed base weights stay frozen, B, its fp32 master and the Adam moments are untouched, so a
        sync between two steps (a periodic checkpoint) does not alter the optimisation, and export_lora_state_dict() keeps returning the
        adapters (the reference saves them un-merged: train.py:962 get_peft_state_maybe_zero_3)."""
        if self.lora_bucket is not None:
            self._write_lora_merged_params()
        if not self.train_decoder:
            return
        own = dict(self.model.named_parameters())
        with torch.no_grad():
            for k, v in self.export_state_dict().items():
                own[k].data.copy_(v.to(own[k].dtype))

    def _lora_targets(self):
        cfg = self.cfg
        H, KV, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        return (("q_proj", 0, H * hd), ("v_proj", (H + KV) * hd, (H + 2 * KV) * hd))

    def _write_lora_merged_params(self):
        """nn.Parameters of the adapted projections <- bf16(W + (alpha / r) B A); reads the trainer's buffers, writes none of them"""
        lb, r, Rp = self.lora_bucket, self.lora_r, self.Rp
        own = dict(self.model.named_parameters())
        with torch.no_grad():
            for i, b in enumerate(self.layers):
                wqkv = b.view(b.w, "wqkv")
                a = lb.view(lb.master, f"Acat.{i}")
                for j, (nm, lo, hi) in enumerate(self._lora_targets()):
                    bm = lb.view(lb.master, f"B{nm[0]}.{i}")
                    delta = (bm[:, :r].float() @ a[j * Rp:j * Rp + r].float()) * self.lora_scale
                    key = f"model.layers.{i}.self_attn.{nm}.weight"
                    own[key].data.copy_((wqkv[lo:hi].float() + delta).to(own[key].dtype))

    def export_lora_state_dict(self):
        """the adapters under peft's names (what the reference saves with get_peft_state_maybe_zero_3, train.py:962): lora_A [r, in], lora_B [out, r], fp32"""
        sd, r, Rp, lb = {}, self.lora_r, self.Rp, self.lora_bucket
        for i in range(len(self.layers)):
            a = lb.view(lb.w, f"Acat.{i}")
            for j, nm in enumerate(("q_proj", "v_proj")):
                p = f"base_model.model.model.layers.{i}.self_attn.{nm}."
                sd[p + "lora_A.weight"] = a[j * Rp:j * Rp + r].clone()
                sd[p + "lora_B.weight"] = lb.view(lb.w, f"B{nm[0]}.{i}")[:, :r].clone()
        return sd

    def merge_lora(self):
        """peft merge_and_unload, explicitly: W <- W + (alpha / r) B A into the PACKED base weights and the model's parameters, B (working copy and fp32
        master) zeroed afterwards so that a second call adds nothing.  This ends the adapters' training run (their Adam moments no longer describe B);
        sync_to_model() / detach() never call it."""
        lb, r, Rp = self.lora_bucket, self.lora_r, self.Rp
        own = dict(self.model.named_parameters())
        with torch.no_grad():
            for i, b in enumerate(self.layers):
                wqkv = b.view(b.w, "wqkv")
                a = lb.view(lb.w, f"Acat.{i}")
                for j, (nm, lo, hi) in enumerate(self._lora_targets()):
                    bm = lb.view(lb.w, f"B{nm[0]}.{i}")
                    delta = (bm[:, :r] @ a[j * Rp:j * Rp + r]) * self.lora_scale
                    wqkv[lo:hi].copy_((wqkv[lo:hi].float() + delta).to(wqkv.dtype))
                    key = f"model.layers.{i}.self_attn.{nm}.weight"
                    own[key].data.copy_(wqkv[lo:hi].to(own[key].dtype))
                    bm.zero_()
            lb.master.copy_(lb.w)
        self._refresh_lora()
        self._refresh_transposes()

    def detach(self):
        """sync_to_model() (with adapters: the merged view W + (alpha / r) B A in the parameters), then release the model: its packed buffers are rebuilt
        from the (now current) parameters on next use"""
        self.sync_to_model()
        if getattr(self, "_splitk_before", None) is not None:
            from . import _lib as _L
            _L.load().ufv_gemm_set_splitk(self._splitk_before)
            self._splitk_before = None
        self.model.get_model()._owner = None
        self.model._owner = None
        self.model.invalidate(); self.model.get_model().invalidate()

    def state_dict(self):
        """This rank's optimizer state: step count, learning rates and, per bucket, the fp32 master / m / v SHARDS (ZeRO-2: each rank
        saves and restores its own slice; load on the same world size)."""
        sd = {"t": self.t, "lr": self.lr, "base_lr": self.base_lr, "mm_projector_lr": self.mm_projector_lr, "world": self.world, "rank": self.rank,
              "buckets": []}
        for b in self.buckets():
            sd["buckets"].append({"n": b.n, "master": b.master.clone(), "m": b.m.clone(), "v": b.v.clone()})
        return sd

    def load_state_dict(self, sd):
        if sd["world"] != self.world or sd["rank"] != self.rank:
            raise ValueError(f"optimizer shards were saved by rank {sd['rank']} of {sd['world']}, this is rank {self.rank} of {self.world}")
        bs = self.buckets()
        if len(bs) != len(sd["buckets"]) or any(b.n != e["n"] for b, e in zip(bs, sd["buckets"])):
            raise ValueError("optimizer state does not match this trainer's bucket layout")
        self.t, self.lr, self.base_lr, self.mm_projector_lr = sd["t"], sd["lr"], sd["base_lr"], sd["mm_projector_lr"]
        for b, e in zip(bs, sd["buckets"]):
            b.master.copy_(e["master"]); b.m.copy_(e["m"]); b.v.copy_(e["v"])
            if b.world == 1:
                b.w.copy_(b.master)
            else:
                lo = b.rank * b.shard
                b.w[lo:lo + b.shard].copy_(b.master)
                all_gather_shards(b.w, self.group)
        if self.proj_bucket is not None:
            if self.proj_flat is not None:
                self.proj_flat.copy_(self.proj_bucket.master)
            else:
                for k, v in self.proj_params.items():
                    v.data.copy_(self.proj_bucket.view(self.proj_bucket.w, k))
            for mod in self.aux_modules.values():
                mod._packed = None
        if self.train_decoder:
            self._refresh_transposes()

    def set_lr_ratio(self, ratio):
        """learning rates of the next step() = initial rates x ratio (what an LR scheduler does to every optimizer group)"""
        self.lr = self.base_lr * ratio
        if self.base_mm_projector_lr is not None:
            self.mm_projector_lr = self.base_mm_projector_lr * ratio

    # ---- export in the reference's parameter names --------------------------------------------------------------------
    def _decoder_names(self):
        """[(reference parameter name, bucket, entry, row slice | 'gate' | 'up' | None)] of the decoder's parameters (HF Qwen2 names)"""
        cfg = self.cfg
        H, KV, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        out = []
        for i, b in enumerate(self.layers):
            p = f"model.layers.{i}."
            for nm, lo, hi in (("q_proj", 0, H * hd), ("k_proj", H * hd, (H + KV) * hd), ("v_proj", (H + KV) * hd, (H + 2 * KV) * hd)):
                out.append((p + f"self_attn.{nm}.weight", b, "wqkv", (lo, hi)))
                out.append((p + f"self_attn.{nm}.bias", self.small, f"bqkv.{i}", (lo, hi)))
            out += [(p + "self_attn.o_proj.weight", b, "wo", None), (p + "mlp.gate_proj.weight", b, "wgu", "gate"), (p + "mlp.up_proj.weight", b, "wgu", "up"),
                    (p + "mlp.down_proj.weight", b, "wd", None), (p + "input_layernorm.weight", self.small, f"ln1.{i}", None),
                    (p + "post_attention_layernorm.weight", self.small, f"ln2.{i}", None)]
        out += [("model.norm.weight", self.small, "norm", None), ("lm_head.weight", self.head, "lm_head", (0, self.V))]
        if self.train_embed:
            out.append(("model.embed_tokens.weight", self.head, "embed", None))
        return out

    def _slice(self, buf_view, sel):
        I = self.cfg.intermediate_size
        if sel is None:
            return buf_view
        if sel == "gate" or sel == "up":                       # gate / up rows are interleaved in blocks of 16 (pack_swiglu)
            return buf_view.view(I // 16, 2, 16, -1)[:, 0 if sel == "gate" else 1]
        return buf_view[sel[0]:sel[1]]

    def export_grad_dict(self, dtype=torch.float32, only=None):
        """d(loss)/d(parameter) of the last accumulation window under the reference's parameter names, un-packed (q / k / v split, gate / up
        de-interleaved), as copies in `dtype`: decoder (when trained), mm_projector / region_encoder / text_hidden_fcs / mask decoder (when trained).
        `only`: a set of names to export.  This rank's gradients BEFORE any exchange (world 1: the gradients)."""
        out = {}
        if self.train_decoder:
            I = self.cfg.intermediate_size
            for name, b, entry, sel in self._decoder_names():
                if only is not None and name not in only:
                    continue
                g = self._slice(b.view(b.g, entry), sel)
                g = g.reshape(I, -1) if sel in ("gate", "up") else g
                out[name] = g.to(dtype, copy=True)
        if self.proj_bucket is not None:
            pb = self.proj_bucket
            for k in self.proj_params:
                if only is None or ("model." + k) in only:
                    out["model." + k] = pb.view(pb.g, k).to(dtype, copy=True)
        return out

    def refresh_from_model(self):
        """The other direction of sync_to_model(): the model's nn.Parameters (just updated by an optimizer of the caller's) -> this engine's packed
        buffers, transposes and the auxiliary modules' packed copies.  Used by the autograd path before every forward whose parameters changed."""
        own = dict(self.model.named_parameters())
        with torch.no_grad():
            for name, b, entry, sel in self._decoder_names():
                dst = self._slice(b.view(b.w, entry), sel)
                src = own[name].data
                dst.copy_(src.view(dst.shape[0], 16, -1) if sel in ("gate", "up") else src)
                if self.optimizer_states and b.master is not None and b.world == 1:
                    self._slice(b.view(b.master, entry), sel).copy_(src.view(dst.shape[0], 16, -1) if sel in ("gate", "up") else src)
            if self.proj_bucket is not None:
                pb = self.proj_bucket
                for k, v in self.proj_params.items():
                    pb.view(pb.w, k).copy_(v.data)
                for mod in self.aux_modules.values():
                    mod.invalidate()
        self._refresh_transposes()

    def export_state_dict(self):
        """Updated decoder weights under the reference's (HF Qwen2) key names, un-packed (q/k/v split, gate/up de-interleaved)."""
        cfg = self.cfg
        H, KV, hd, I = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.intermediate_size
        sd = {}
        for i, b in enumerate(self.layers):
            p = f"model.layers.{i}."
            wqkv = b.view(b.w, "wqkv")
            bq = self.small.view(self.small.w, f"bqkv.{i}")
            for nm, lo, hi in (("q_proj", 0, H * hd), ("k_proj", H * hd, (H + KV) * hd), ("v_proj", (H + KV) * hd, (H + 2 * KV) * hd)):
                sd[p + f"self_attn.{nm}.weight"] = wqkv[lo:hi].clone()
                sd[p + f"self_attn.{nm}.bias"] = bq[lo:hi].to(torch.bfloat16)
            sd[p + "self_attn.o_proj.weight"] = b.view(b.w, "wo").clone()
            gu = b.view(b.w, "wgu").view(I // 16, 2, 16, -1)
            sd[p + "mlp.gate_proj.weight"] = gu[:, 0].reshape(I, -1).clone()
            sd[p + "mlp.up_proj.weight"] = gu[:, 1].reshape(I, -1).clone()
            sd[p + "mlp.down_proj.weight"] = b.view(b.w, "wd").clone()
            sd[p + "input_layernorm.weight"] = self.small.view(self.small.w, f"ln1.{i}").to(torch.bfloat16)
            sd[p + "post_attention_layernorm.weight"] = self.small.view(self.small.w, f"ln2.{i}").to(torch.bfloat16)
        sd["model.norm.weight"] = self.small.view(self.small.w, "norm").to(torch.bfloat16)
        sd["lm_head.weight"] = self.head.view(self.head.w, "lm_head")[:self.V].clone()
        if self.train_embed:
            sd["model.embed_tokens.weight"] = self.head.view(self.head.w, "embed").clone()
        return sd


def warmup_cosine_ratio(it, total_num_steps, warmup_num_steps, warmup_min_ratio=0.0, cos_min_ratio=0.03, warmup_type="linear"):
    """LR multiplier at scheduler iteration `it` (0-based count of optimizer steps already taken) of the schedule the reference
    trains with: DeepSpeed `WarmupCosineLR` as configured in scripts/zero2.json:13-22 (warmup_min_ratio 0, cos_min_ratio 0.03,
    linear warm-up; total / warm-up steps filled in by the HF Trainer).  DeepSpeed (0.17.5 pinned, requirements.txt:33) is not in
    this image and the reference holds no test for it: restated from its published lr_schedules.py -- parity unpinned.
      it <  warmup:  warmup_min_ratio + (1 - warmup_min_ratio) * (it / warmup   |   log(it + 1) / log(warmup))
      it >= warmup:  max(0, cos_min_ratio + (1 - cos_min_ratio) * (1 + cos(pi * (it - warmup + 1) / (total - warmup))) / 2)"""
    warmup = max(2, int(warmup_num_steps))
    if it < 0:
        return 0.0
    if it < warmup:
        r = it / warmup if warmup_type == "linear" else math.log(it + 1) / math.log(warmup)
        return warmup_min_ratio + (1.0 - warmup_min_ratio) * r
    done, span = it - warmup + 1, max(1, total_num_steps - warmup)
    return max(0.0, cos_min_ratio + (1.0 - cos_min_ratio) * (1.0 + math.cos(math.pi * done / span)) / 2.0)


def shard_bounds(n, world, rank):
    """[lo, hi) of rank's shard of a flat buffer of n elements (n a multiple of world): the ZeRO-2 partition"""
    assert n % world == 0
    s = n // world
    return rank * s, (rank + 1) * s


def zero2_step_reference(params, grad, states, lr, betas, eps, wd, t, max_grad_norm, group=None):
    """The same ZeRO-2 exchange as DecoderTrainer.step() (reduce_scatter_mean -> global-norm clip -> AdamW on the local shard
    -> all_gather_shards) with the update written in plain torch, so that it runs on any device / backend: the gloo
    world_size-2 test pins the partition arithmetic and the exchange order without a GPU.  params flat fp32 [n]
    (replicated, updated in place), grad this rank's flat gradient, states = (m, v) shards."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(params.numel(), world, rank)
    gsh = reduce_scatter_mean(torch.empty((hi - lo,), dtype=params.dtype), grad.contiguous(), group)
    sq = (gsh * gsh).sum()
    dist.all_reduce(sq, group=group)
    coef = torch.clamp(max_grad_norm / (sq.sqrt() + 1e-6), max=1.0) if max_grad_norm else torch.tensor(1.0)
    g = gsh * coef
    m, v = states
    b1, b2 = betas
    p = params[lo:hi]
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    p.addcdiv_(m, v.sqrt() / math.sqrt(bc2) + eps, value=-lr / bc1)
    all_gather_shards(params, group)
    return params
