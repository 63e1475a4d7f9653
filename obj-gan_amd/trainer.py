"""G+D training driver of the image generator on MI355X.

Same entry points as the reference (reference image_generation/trainer.py:30-510):
`condGANTrainer(output_dir, data_loader, dataset)` with `build_models`, `define_optimizers`,
`prepare_labels`, `save_model`, `train`.  The per-iteration body of the reference's loop
(trainer.py:357-472) is `train_step`, usable on its own (bench.py, tests).

MI355X-first differences (results-preserving unless noted):
  * every network's parameters, gradients and Adam moments live in flat fp32 arenas
    (`ParamArena`): one fused Adam launch per network, one fused EMA launch, and -- under
    data parallelism -- ONE RCCL all-reduce per network on the flat gradient arena;
  * data parallelism is one process per GPU over torch.distributed (backend "nccl" = RCCL over
    xGMI) instead of the reference's single-process nn.DataParallel (trainer.py:136-152): each
    rank runs the reference step on its own minibatch shard (per-rank BatchNorm statistics and
    DAMSM negatives, exactly like a DataParallel replica would for BN), gradients are averaged.
    The all-reduce of discriminator i is asynchronous and overlaps the forward/backward of
    discriminator i+1; Adam for the discriminators runs once their reductions have landed;
  * discriminator weights are frozen (requires_grad=False) during the generator step: the
    reference computes and discards those weight gradients (trainer.py:449; SURVEY.md trap 7);
  * one device->host read of num_rois per step instead of one per stage; log strings are only
    formatted on print steps (the reference calls .item() on every loss every step).
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from miscc.config import cfg
from miscc.utils import (mkdir_p, weights_init, form_clabels_feat, _host, compute_inception_score, draw_class_permutations,
                         negative_log_posterior_probability)
from miscc.losses import patD_loss, shpD_loss, objD_loss, G_loss, KL_loss, patD_real, shpD_real, objD_real
import model as M
from model import (G_NET, PAT_D_NET64, PAT_D_NET128, PAT_D_NET256, SHP_D_NET64, SHP_D_NET128,
                   SHP_D_NET256, OBJ_SS_D_NET, OBJ_LS_D_NET)
from objgan_hip import ops


# ---------------------------------------------------------------------------------------------
# flat parameter / gradient / moment arenas
# ---------------------------------------------------------------------------------------------
class ParamArena(object):
    """Re-homes a module's parameters into one flat fp32 buffer (and their .grad into another).

    grad has one extra trailing element: the "this rank contributed" flag used to keep the
    conditional object-discriminator updates consistent across ranks."""

    def __init__(self, module):
        self.module = module
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.n = n
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = 0
        # device-side step state of the gated update (ArenaAdam.step(gated=True)): steps, beta1^steps,
        # beta2^steps as float64 + 3 floats of kernel scratch
        self.gate_state = None
        self.gate_coef = None
        self._views = []
        self.epoch = [0]      # bumped by every optimizer step: ops' packed-filter cache keys on it
        off = 0
        for p in self.params:
            p._og_epoch = self.epoch
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            gview = self.grad[off:off + k].view_as(p)
            p.grad = gview
            # ops' convolution backward accumulates a weight gradient straight into this view and reports here in
            # AccumulateGrad's place (ops._grad_sink)
            p._og_grad_sink = (gview, lambda i=len(self._views): self._mark(i, True))
            self._views.append(gview)
            off += k

    _armed = None

    def zero_grad(self):
        self.grad.zero_()
        for p, gv in zip(self.params, self._views):
            p.grad = gv

    # ---- gradient-ready notification, in buckets (data-parallel overlap of the generator's all-reduce) ----
    def make_buckets(self, k=4):
        """Split the arena into <= k contiguous ranges of whole parameters (registration order, about equal
        sizes) and install the per-parameter 'gradient is final' notification (AccumulateGrad's post-hook)."""
        if getattr(self, "buckets", None) is not None:
            return self.buckets
        target = max(1, self.n // k)
        self.buckets, self._bucket_of = [], []
        start = off = 0
        members = []
        for i, p in enumerate(self.params):
            members.append(i)
            off += p.numel()
            if off - start >= target and len(self.buckets) < k - 1:
                self.buckets.append((start, off, members))
                start, members = off, []
        if members:
            self.buckets.append((start, off, members))
        self._bucket_of = [0] * len(self.params)
        for b, (_, _, mem) in enumerate(self.buckets):
            for i in mem:
                self._bucket_of[i] = b
        self._armed = None
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(lambda *_a, i=i: self._mark(i))
            p.register_hook(lambda g, i=i: self._autograd_piece(i, g))   # (g is None when autograd was handed no gradient)
        return self.buckets

    def arm(self, on_bucket):
        """Until disarm(): on_bucket(start, end) is called once per bucket, as soon as every parameter of the
        bucket has reported its gradient (each parameter of the network is used once per forward) AND every
        higher bucket has been handed over: buckets leave in descending order, always.  The backward pass
        completes them in that order anyway (the last layers first); pinning it makes the sequence of
        collectives identical on every rank even when a rank's batch prunes a branch of its graph."""
        self.make_buckets()
        self._armed = (on_bucket, [set(mem) for (_, _, mem) in self.buckets], [len(self.buckets) - 1])
        self._sink_marked, self._hook_marked = set(), set()

    def _two_pieces(self, i):
        names = [k for k, p_ in self.module.named_parameters() if p_.requires_grad]
        raise RuntimeError("ParamArena: parameter %d (%s) is used more than once per forward (its gradient arrived "
                           "in two pieces during one armed backward pass): the bucketed all-reduce would ship a "
                           "partial gradient -- switch the trainer's direct_wgrad off for this network"
                           % (i, names[i] if i < len(names) else "?"))

    def _autograd_piece(self, i, g):
        """tensor hook of parameter i: autograd is about to accumulate the gradient g into it (the hook also fires with
        g = None when the backward function handed autograd no gradient -- the sink's case)"""
        if g is not None and self._armed is not None and i in self._sink_marked:
            self._two_pieces(i)            # the sink already reported this parameter as complete
        return None

    def _mark(self, i, from_sink=False):
        if self._armed is None:
            return
        # AccumulateGrad's post-hook fires once per backward pass and parameter, after ALL contributions that went through
        # autograd were summed (also when the only contribution went through the sink and autograd was handed None).  The
        # direct weight-gradient sink of ops (`_grad_sink`) reports once per USE of the weight, from inside that use's
        # backward.  A weight used twice per forward -- two sink reports, a sink report after the post-hook, or a sink
        # report followed by a gradient through autograd (`_autograd_piece`) -- would hand its bucket to the all-reduce
        # before its gradient is complete.  Refuse loudly.
        if from_sink:
            if i in self._sink_marked or i in self._hook_marked:
                self._two_pieces(i)
            self._sink_marked.add(i)
        else:
            self._hook_marked.add(i)
        on_bucket, waiting, nxt = self._armed
        waiting[self._bucket_of[i]].discard(i)
        while nxt[0] >= 0 and not waiting[nxt[0]]:
            b = nxt[0]
            nxt[0] -= 1
            on_bucket(self.buckets[b][0], self.buckets[b][1])

    def disarm(self):
        """-> the (start, end) ranges that were not handed over during the backward pass (a parameter
        without a gradient this step holds its bucket and the lower ones back), in the same descending order."""
        if self._armed is None:
            return []
        _, _, nxt = self._armed
        self._armed = None
        return [self.buckets[b][:2] for b in range(nxt[0], -1, -1)]

    def sync_grads(self):
        """Make sure every gradient lives in the arena (callers may have used
        module.zero_grad(set_to_none=True) or assigned .grad themselves)."""
        for p, gv in zip(self.params, self._views):
            if p.grad is None:
                gv.zero_()
                p.grad = gv
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
                p.grad = gv

    def set_requires_grad(self, flag):
        for p in self.params:
            p.requires_grad_(flag)


class ArenaAdam(object):
    """torch.optim.Adam(params, lr, betas) semantics on a ParamArena: one fused launch."""

    def __init__(self, arena, lr, betas=(0.5, 0.999), eps=1e-8):
        self.arena = arena
        self.lr, self.betas, self.eps = lr, betas, eps
        self.param_groups = [{"params": arena.params, "lr": lr, "betas": betas, "eps": eps}]

    def zero_grad(self):
        self.arena.zero_grad()

    def step(self, grad_scale=1.0, gated=False):
        """gated=True: the update is taken only if the flag slot behind the gradient arena
        (arena.grad[-1], summed over ranks by the gradient all-reduce) is positive -- decided on the
        device, with the step counter on the device too: no host sync (reference trainer.py:429,440
        tests `float(err) > 0` on the host)."""
        a = self.arena
        if hasattr(ops, "wgrad_join"):
            ops.wgrad_join()                    # weight gradients that were handed to a side stream (ops.wgrad_stream_scope)
        a.sync_grads()
        a.epoch[0] += 1
        lr = self.param_groups[0]["lr"]
        if gated:
            if a.gate_state is None:
                a.gate_state = torch.tensor([float(a.step_count), self.betas[0] ** a.step_count,
                                             self.betas[1] ** a.step_count], dtype=torch.float64,
                                            device=a.flat.device)
                a.gate_coef = torch.zeros(3, dtype=torch.float32, device=a.flat.device)
            ops.adam_step_gated_(a.flat, a.grad, a.exp_avg, a.exp_avg_sq, lr, self.betas[0], self.betas[1],
                                 self.eps, a.gate_state, a.grad[a.n:], a.gate_coef, grad_scale=grad_scale,
                                 n=a.n)
            ops.repack_arena(a.epoch)
            return
        if a.gate_state is not None:        # leave gated mode: one read of the device counter
            a.step_count = int(a.gate_state[0].item())
            a.gate_state = None
        a.step_count += 1
        ops.adam_step_(a.flat, a.grad, a.exp_avg, a.exp_avg_sq, lr,
                       self.betas[0], self.betas[1], self.eps, a.step_count, grad_scale=grad_scale,
                       n=a.n)
        ops.repack_arena(a.epoch)       # every cached filter bank of this network, one launch

    @property
    def steps_taken(self):
        a = self.arena
        return int(a.gate_state[0].item()) if a.gate_state is not None else a.step_count

    def state_dict(self):
        return {"step": self.steps_taken, "exp_avg": self.arena.exp_avg,
                "exp_avg_sq": self.arena.exp_avg_sq}


def category_embeddings(glove_weight, cat_labels, cat_label_lens, sorted_cat_label_indices, num_cats):
    """Mean GloVe vector of one- and two-word category names; rows of longer names stay zero and
    only the first `num_cats` rows are filled (both as in reference trainer.py:226-242).
    `cat_labels` / `cat_label_lens` are sorted by name length, `sorted_cat_label_indices` is the
    permutation back to file order (miscc/load.py load_cat_label)."""
    w = glove_weight.detach()
    labels = cat_labels.to(w.device)
    lens = cat_label_lens.to(w.device)[:labels.size(0)]
    raw = w[labels.reshape(-1)].view(labels.size(0), labels.size(1), -1)
    first = raw[:, 0]
    second = raw[:, 1] if raw.size(1) > 1 else torch.zeros_like(first)
    out = torch.where((lens == 1).unsqueeze(1), first, torch.zeros_like(first))
    out = torch.where((lens == 2).unsqueeze(1), (first + second) / 2., out)
    out[num_cats:] = 0
    return out[sorted_cat_label_indices.to(w.device)]


class _NullCtx(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _dist_on():
    """A process group exists: the data-parallel path is taken, at any world size (a one-rank group runs
    the same collectives -- that is how the RCCL path is exercised on a single-GPU box)."""
    return dist.is_available() and dist.is_initialized()


# ---------------------------------------------------------------------------------------------
# trainer
# ---------------------------------------------------------------------------------------------
def resolve_d_job_order(job_names, order, rng_free=False):
    """The issue order of the discriminator jobs of one step under a `d_job_order` / OBJGAN_D_ORDER request: the listed jobs
    first, the rest behind them in the reference order (reference trainer.py:398-443).  Every job but the patch
    discriminators draws from the python RNG (permute_seg, miscc/utils.py), so the FINAL order must keep those jobs in the
    reference sequence -- a partial list such as ["errShpD1"] would otherwise run ShpD1 ahead of ShpD0 and silently lose the
    reference parity of the step.  Unknown or repeated names raise."""
    order = list(order)
    unknown = [n for n in order if n not in job_names]
    if unknown or len(set(order)) != len(order):
        raise ValueError("d_job_order names jobs that do not exist in this step (or names one twice): %s; jobs: %s"
                         % (unknown or order, list(job_names)))
    final = order + [n for n in job_names if n not in order]
    ref = [n for n in job_names if not n.startswith("errPatD")]
    # rng_free: the permutations were drawn up front in the reference order (train_step) -- any issue order is safe
    if not rng_free and [n for n in final if not n.startswith("errPatD")] != ref:
        raise ValueError("d_job_order moves a job that draws random numbers out of the reference order: %s "
                         "(issue order would be %s)" % (order, final))
    return final


class condGANTrainer(object):
    def __init__(self, output_dir, data_loader, dataset, device=None):
        if cfg.TRAIN.FLAG and output_dir:
            self.model_dir = os.path.join(output_dir, 'Model')
            self.image_dir = os.path.join(output_dir, 'Image')
            self.score_dir = os.path.join(output_dir, 'Score')
            for d in (self.model_dir, self.image_dir, self.score_dir):
                mkdir_p(d)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.batch_size = cfg.TRAIN.BATCH_SIZE
        self.max_epoch = cfg.TRAIN.MAX_EPOCH
        self.snapshot_interval = cfg.TRAIN.SNAPSHOT_INTERVAL
        self.print_interval = cfg.TRAIN.PRINT_INTERVAL
        self.data_loader = data_loader
        self.dataset = dataset
        self.num_batches = len(data_loader) if data_loader is not None else 0
        self.num_classes = getattr(dataset, "num_classes", None) or len(getattr(dataset, "cats_index_dict", range(80)))
        self.text_encoder = getattr(dataset, "text_encoder", None)
        self.image_encoder = getattr(dataset, "image_encoder", None)
        self.inception_model = getattr(dataset, "inception_model", None)
        self.glove_emb = getattr(dataset, "glove_embed", None)       # nn.Embedding(n, 50), frozen
        # category-name tables of a real dataset (reference trainer.py:50-55); absent for synthetic batches
        for name in ("n_words", "ixtoword", "cats_index_dict", "cat_labels", "cat_label_lens",
                     "sorted_cat_label_indices"):
            setattr(self, name, getattr(dataset, name, None))
        self.ddp = _dist_on()
        self.rank = dist.get_rank() if self.ddp else 0
        self.world = dist.get_world_size() if self.ddp else 1
        self.is_stream = torch.cuda.Stream(device=self.device) \
            if self.inception_model is not None and self.device.type == "cuda" else None
        # The object discriminators consume the second stage's bottom-up codes: a one-stage tree cannot
        # have them (the reference raises IndexError there, SURVEY.md trap 8); `use_obj = False` also
        # gives BASELINE.json config 3 "without the two object discriminators".
        self.use_obj = cfg.TREE.BRANCH_NUM >= 2

    # ---- model / optimizer construction ---------------------------------------------------------
    def build_models(self):
        """-> [text_encoder, image_encoder, netG, netsPatD, netsShpD, netObjSSD, netObjLSD, epoch]
        (reference trainer.py:75-195).  The frozen encoders come from the dataset object (they
        are outside the kernel scope of this round); checkpoints named like the reference's
        (netG_epoch_%d.pth, netPatD%d.pth, ...) are loaded when cfg.TRAIN.NET_G is set."""
        self._build_encoders()
        nc = self.num_classes
        netG = G_NET(nc)
        netsPatD, netsShpD = [], []
        pat_cls = (PAT_D_NET64, PAT_D_NET128, PAT_D_NET256)
        shp_cls = (SHP_D_NET64, SHP_D_NET128, SHP_D_NET256)
        for i in range(cfg.TREE.BRANCH_NUM):
            netsPatD.append(pat_cls[i]())
            netsShpD.append(shp_cls[i](nc))
        netObjSSD, netObjLSD = OBJ_SS_D_NET(nc), OBJ_LS_D_NET(nc)
        nets = [netG] + netsPatD + netsShpD + [netObjSSD, netObjLSD]
        for net in nets:
            net.apply(weights_init)
        epoch = 0
        if cfg.TRAIN.NET_G != '':
            sd = torch.load(cfg.TRAIN.NET_G, map_location="cpu")
            netG.load_state_dict(sd)
            name = cfg.TRAIN.NET_G
            epoch = int(name[name.rfind('_') + 1:name.rfind('.')]) + 1
            base = name[:name.rfind('/')]
            for i, d in enumerate(netsPatD):
                d.load_state_dict(torch.load('%s/netPatD%d.pth' % (base, i), map_location="cpu"))
            for i, d in enumerate(netsShpD):
                d.load_state_dict(torch.load('%s/netShpD%d.pth' % (base, i), map_location="cpu"))
            netObjSSD.load_state_dict(torch.load('%s/netObjSSD.pth' % base, map_location="cpu"))
            netObjLSD.load_state_dict(torch.load('%s/netObjLSD.pth' % base, map_location="cpu"))
        for net in nets:
            net.to(self.device).train()
        if self.ddp:            # identical replicas: rank 0's weights everywhere
            for net in nets:
                for t in list(net.parameters()) + list(net.buffers()):
                    dist.broadcast(t.data, src=0)
        return [self.text_encoder, self.image_encoder, netG, netsPatD, netsShpD, netObjSSD,
                netObjLSD, epoch]

    def _build_encoders(self):
        """The frozen encoders of the step (reference trainer.py:62-100): caption encoder from
        cfg.TRAIN.NET_E, image encoder from the sibling `image_encoder*.pth`, and the Inception-v3 of the
        per-step Inception-score monitor.  A dataset object may hand in ready-made ones (bench, tests);
        otherwise they are built and loaded here exactly as the reference does -- the monitor's ImageNet
        weights, which the reference downloads, are read from cfg.TRAIN.NET_INCEPTION or the torch hub
        cache (no network here): without that file the monitor is off."""
        import encoders
        from model import RNN_ENCODER
        # a prepared data set (it has a vocabulary) gets its encoders here; synthetic batches may carry
        # precomputed caption embeddings / their own encoder objects instead
        real = self.n_words is not None
        need_text = real and self.text_encoder is None
        need_img = real and self.image_encoder is None
        if (need_text or need_img) and cfg.TRAIN.NET_E == '':
            raise RuntimeError('Error: no pretrained text-image encoders (cfg.TRAIN.NET_E)')
        if need_img:
            path = cfg.TRAIN.NET_E.replace('text_encoder', 'image_encoder')
            enc = encoders.CNN_ENCODER(cfg.TEXT.EMBEDDING_DIM)
            enc.load_state_dict(torch.load(path, map_location="cpu"))
            print('Load image encoder from:', path)
            self.image_encoder = enc
        if need_text:
            enc = RNN_ENCODER(self.n_words, nhidden=cfg.TEXT.EMBEDDING_DIM)
            enc.load_state_dict(torch.load(cfg.TRAIN.NET_E, map_location="cpu"))
            print('Load text encoder from:', cfg.TRAIN.NET_E)
            self.text_encoder = enc
        if real and self.inception_model is None:
            path = getattr(cfg.TRAIN, "NET_INCEPTION", '') or os.path.join(
                os.environ.get("TORCH_HOME", os.path.expanduser("~/.cache/torch")), "hub", "checkpoints",
                "inception_v3_google-1a9a5a14.pth")
            if os.path.exists(path):
                net = encoders.inception_v3()
                net.load_state_dict(torch.load(path, map_location="cpu"))
                self.inception_model = encoders.INCEPTION_V3(net)
            elif self.rank == 0:
                print('Inception-score monitor off: no ImageNet weights at', path)
        for enc in (self.text_encoder, self.image_encoder, self.inception_model):
            if isinstance(enc, nn.Module):
                for p in enc.parameters():
                    p.requires_grad_(False)
                enc.to(self.device).eval()
        if self.inception_model is not None and self.is_stream is None and self.device.type == "cuda":
            self.is_stream = torch.cuda.Stream(device=self.device)

    def define_optimizers(self, netG, netsPatD, netsShpD, netObjSSD, netObjLSD):
        d_lr, g_lr = cfg.TRAIN.DISCRIMINATOR_LR, cfg.TRAIN.GENERATOR_LR
        optimizersPatD = [ArenaAdam(ParamArena(d), d_lr) for d in netsPatD]
        optimizersShpD = [ArenaAdam(ParamArena(d), d_lr) for d in netsShpD]
        optimizerObjSSD = ArenaAdam(ParamArena(netObjSSD), d_lr)
        optimizerObjLSD = ArenaAdam(ParamArena(netObjLSD), d_lr)
        optimizerG = ArenaAdam(ParamArena(netG), g_lr)
        return optimizerG, optimizersPatD, optimizersShpD, optimizerObjSSD, optimizerObjLSD

    def prepare_labels(self):
        return torch.arange(self.batch_size, dtype=torch.long, device=self.device)

    def prepare_cat_emb(self):
        """[num categories, 50] GloVe embedding of every category name, in categories.txt order
        (reference trainer.py:226-242)."""
        return category_embeddings(self.glove_emb.weight, self.cat_labels, self.cat_label_lens,
                                   self.sorted_cat_label_indices, len(self.cats_index_dict)).to(self.device)

    def setup(self):
        """Build networks, arenas, the EMA copy and the per-run constants."""
        (self.text_encoder, self.image_encoder, self.netG, self.netsPatD, self.netsShpD,
         self.netObjSSD, self.netObjLSD, self.start_epoch) = self.build_models()
        (self.optimizerG, self.optimizersPatD, self.optimizersShpD, self.optimizerObjSSD,
         self.optimizerObjLSD) = self.define_optimizers(self.netG, self.netsPatD, self.netsShpD,
                                                        self.netObjSSD, self.netObjLSD)
        self.avg_param_G = self.optimizerG.arena.flat.clone()      # EMA of G, flat
        if self.glove_emb is not None:
            self.glove_emb.to(self.device).eval()
        self.clabels_emb = self.prepare_cat_emb() if self.cat_labels is not None else None
        self.match_labels = self.prepare_labels()
        self.noise = torch.empty(self.batch_size, cfg.GAN.Z_DIM, device=self.device)
        self.gen_iterations = 0
        return self

    # The frozen Inception chains (DAMSM image encoder: forward + backward w.r.t. the image; Inception-score monitor:
    # forward) are shape-static and launch-bound: each is captured once into hipGraphs and replayed (objgan_hip.graphs).
    use_graphs = True
    # the discriminators' real-image passes run on the side streams beside the generator's forward pass (train_step (1b))
    # (measured in round 6: 136.7 / 133.9 ms with, 135.0 / 134.5 without on one box -- the device has no idle capacity left
    # during the generator's forward pass; off by default, profiles/r06_ab_variants.txt)
    hoist_real_passes = os.environ.get("OBJGAN_HOIST_REAL", "0") == "1"
    # fp16x2 run-time guard: every `h2_guard_every` iterations (and on the first) one step is a checked step (0: never)
    h2_guard_every = int(os.environ.get("OBJGAN_H2_GUARD_EVERY", "500"))     # (profiling recipes set 0: a checked step syncs)

    def _graphed(self, attr):
        import encoders
        enc = getattr(self, attr)
        if (not self.use_graphs or self.device.type != "cuda"
                or not isinstance(enc, (encoders.CNN_ENCODER, encoders.INCEPTION_V3))):
            return enc
        cache = self.__dict__.setdefault("_graph_wrappers", {})
        got = cache.get(attr)
        if got is None or got[0] is not enc:
            from objgan_hip import graphs
            fn = enc if attr == "image_encoder" else (lambda x, enc=enc: (enc(x),))
            ver = graphs.graphed_encoder(enc).version
            got = cache[attr] = (enc, graphs.GraphedCallable(fn, version=ver, name=attr))
        return got[1]

    # weight gradients of the generator's backward pass on a side stream (OBJGAN_ASYNC_WGRAD=0: on the issuing stream)
    async_wgrad = os.environ.get("OBJGAN_ASYNC_WGRAD", "1") != "0"

    # the DAMSM term of the generator loss (frozen Inception encoder on the fake image: ~330 small launches, one graph
    # replay) does not depend on the discriminator updates: issue it on its own stream right after the generator's
    # forward pass, beside them.  Same kernels, same autograd sequence (it was already the first term of G_loss):
    # bit-identical steps (tests/test_modules_gpu.py).  OFF: measured 3-4 ms per step SLOWER in steady state (LAB 10.9).
    early_damsm = os.environ.get("OBJGAN_EARLY_DAMSM", "0") == "1"
    async_wgrad_d = os.environ.get("OBJGAN_ASYNC_WGRAD_D", "0") == "1"     # the same inside the discriminator updates (A/B: see LAB)

    def _wgrad_side_stream(self, k=0):
        if (not self.async_wgrad or self.device.type != "cuda" or int(self.d_streams) <= 1 or not self.direct_wgrad
                or not hasattr(ops, "wgrad_stream_scope")):
            return None
        pool = self.__dict__.setdefault("_wg_streams", {})
        st = pool.get(k)
        if st is None:
            st = pool[k] = torch.cuda.Stream(device=self.device)
        return st

    # phase marks of a step on the main stream (measurement aid, off unless `phase_events` is a list: bench.py's host probe)
    phase_events = None

    def _phase(self, name):
        if self.phase_events is not None and self.device.type == "cuda":
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.phase_events.append((name, ev))

    def _d_optimizers(self):
        return self.optimizersPatD + self.optimizersShpD + [self.optimizerObjSSD, self.optimizerObjLSD]

    # ---- concurrency of the discriminator updates ---------------------------------------------------
    # Side streams for the eight discriminator updates and the nine generator-loss terms (1: everything on the
    # caller's stream).  r03, one MI355X, B = 16: 205.6 / 192.7 / 190.0 / 198.5 ms per step with 1 / 2 / 3 / 4 streams,
    # jobs round-robin.  r04 (fp16x2): 148.3 / 142.8 / 155.8 / 141.8 with 2 / 3 / 4 / 5 round-robin -- what matters is
    # which jobs share a stream: the four heavy ones (the 256-px patch and shape discriminators, the two object
    # discriminators) each on a stream of their own and the light ones spread behind them, 5 streams: 139.5
    # (same box, same call: 141.5 with the round-robin three; profiles/r04_ab_rejected_variants.txt (e)).  The stream
    # assignment changes the interleaving on the device, not the arithmetic (bit-identical, tested).
    d_streams = 5
    # side stream of each discriminator job BY NAME (a job without an entry -- another BRANCH_NUM, no object
    # discriminators -- takes the next stream round-robin); OBJGAN_D_STREAM_MAP="0,1,2,0,1,2,0,1" (job order PatD0-2,
    # ShpD0-2, ObjSSD, ObjLSD) overrides it for experiments.
    _D_JOB_NAMES = ("errPatD0", "errPatD1", "errPatD2", "errShpD0", "errShpD1", "errShpD2", "errObjSSD", "errObjLSD")
    d_stream_map = dict(zip(_D_JOB_NAMES, (2, 4, 3, 3, 4, 2, 1, 0)))
    direct_wgrad = True             # weight gradients straight into the arenas during train_step (ops._grad_sink)
    debug_after_d_updates = None    # tests: callable(trainer) between the discriminator updates and the generator loss
    # Order in which the host ISSUES the discriminator jobs (names; jobs not listed keep their reference position behind the
    # listed ones).  The patch discriminators draw no random numbers, so they can move without changing the python RNG
    # sequence of the shape / object discriminators' permute_seg (which stay in the reference's relative order); the
    # networks are independent, so the results do not depend on the order.  OBJGAN_D_ORDER="errPatD2,..." overrides.
    d_job_order = None
    predraw_permutations = True

    @classmethod
    def _stream_map_from_env(cls):
        """the OBJGAN_D_STREAM_MAP override, validated (a malformed value raises here with a message, not as a bare
        ValueError while the class body is executed at import)"""
        raw = os.environ.get("OBJGAN_D_STREAM_MAP")
        if not raw:
            return None
        try:
            vals = [int(v) for v in raw.split(",")]
        except ValueError:
            raise ValueError("OBJGAN_D_STREAM_MAP=%r: expected comma-separated stream indices" % raw)
        if any(v < 0 for v in vals):
            raise ValueError("OBJGAN_D_STREAM_MAP=%r: negative stream index" % raw)
        return dict(zip(cls._D_JOB_NAMES, vals))

    def _d_side_streams(self):
        n = int(self.d_streams)
        if n <= 1 or self.device.type != "cuda":
            return []
        cur = getattr(self, "_d_side", None)
        if cur is None or len(cur) != n:
            self._d_side = cur = [torch.cuda.Stream(device=self.device) for _ in range(n)]
        return cur

    # ---- gradient exchange ------------------------------------------------------------------------
    def _reduce_async(self, opt):
        """One RCCL all-reduce (sum) of a network's flat gradient arena; averaging is folded
        into the Adam kernel's grad_scale."""
        if not self.ddp:
            return None
        opt.arena.sync_grads()
        self._comm_note(opt.arena.grad)
        return dist.all_reduce(opt.arena.grad, op=dist.ReduceOp.SUM, async_op=True)

    # Optional bookkeeping of the gradient exchange (bench.py's `comm` object; off by default): bytes handed to
    # all-reduce, number of collectives, and -- on the GPU -- events around every stream-side wait for a collective, i.e.
    # the time the COMPUTE stream stood still for communication.
    comm = None

    def enable_comm_stats(self, on=True):
        self.comm = {"bytes": 0, "collectives": 0, "waits": []} if on else None

    def _comm_note(self, t):
        if self.comm is not None:
            self.comm["bytes"] += t.numel() * t.element_size()
            self.comm["collectives"] += 1

    def _wait(self, handle):
        """handle.wait() = the current stream waits for the collective (the host runs on)."""
        if self.comm is None or not torch.cuda.is_available() or self.device.type != "cuda":
            handle.wait()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        handle.wait()
        e1.record()
        self.comm["waits"].append((e0, e1))

    def comm_summary(self, steps):
        """-> dict (call after a device synchronize): per-step bytes / collectives / exposed wait on the compute stream"""
        c = self.comm or {"bytes": 0, "collectives": 0, "waits": []}
        steps = max(1, steps)
        return {"allreduce_bytes_per_step": c["bytes"] // steps, "collectives_per_step": c["collectives"] / steps,
                "exposed_wait_ms_per_step": (round(sum(a.elapsed_time(b) for a, b in c["waits"]) / steps, 3)
                                             if c["waits"] else None)}

    # ---- one iteration (reference trainer.py:357-472) -----------------------------------------------
    def train_step(self, batch, noise=None, want_logs=False):
        """batch: dict with imgs[3], hmaps[3], rois[3], fm_rois, num_rois, bt_masks[2], fm_bt_masks,
        words_embs, sent_emb, glove_words_embs, mask, clabels_emb, cap_lens, class_ids (see
        synth_batch.py).  Returns a dict of loss tensors (no host sync unless want_logs)."""
        # weight gradients go straight into the optimizer arenas (ops._grad_sink) for the duration of the step only: under
        # `.backward()` into `.grad`, every arena weight used once per forward.  Anything else (torch.autograd.grad, a
        # weight shared between two uses) keeps autograd's own accumulation.
        guard = (self.h2_guard_every > 0 and self.gen_iterations % self.h2_guard_every == 0
                 and self.device.type == "cuda" and hasattr(ops, "h2_guard_begin") and ops.get_conv_math() == "fp16x2")
        if guard:
            # fp16x2 run-time guard: this step also measures the per-channel spread of every fp16x2 operand whose channels
            # are GEMM rows / columns; a call site beyond 2^16 runs bf16x3 from now on (ops._H2_GUARD; logged)
            ops.h2_guard_begin()
        try:
            with M.deferred_bn_counters(), ops.direct_wgrad_scope(self.direct_wgrad):
                return self._train_step(batch, noise, want_logs)
        finally:
            if guard:
                for rec in ops.h2_guard_end():
                    print("fp16x2 guard, iteration %d: %s (spread 2^%.1f, operand %s) -> bf16x3 from now on"
                          % (self.gen_iterations, rec[1], rec[2], rec[3]))

    def _train_step(self, batch, noise, want_logs):
        b = batch
        imgs, hmaps, rois = b["imgs"], b["hmaps"], b["rois"]
        fm_rois, num_rois = b["fm_rois"], b["num_rois"]
        clabels_emb = b["clabels_emb"]
        # (1) text-side inputs (reference trainer.py:367-380): frozen caption encoder and GloVe table
        # when the dataset provides them; a batch may also carry precomputed embeddings (unit tests)
        mask, glove_words_embs = b.get("mask"), b.get("glove_words_embs")
        if self.text_encoder is not None and "captions" in b:
            captions, cap_lens = b["captions"], b["cap_lens"]
            max_len = int(b.get("max_len", captions.shape[1]))
            with torch.no_grad():
                words_embs, sent_emb = self.text_encoder(captions, cap_lens, max_len)
            num_words = words_embs.size(2)
            mask = (captions == 0)[:, :num_words]
            if self.glove_emb is not None and "glove_captions" in b:
                gc = b["glove_captions"]
                with torch.no_grad():
                    gw = torch.nn.functional.embedding(gc.reshape(-1), self.glove_emb.weight)
                glove_words_embs = gw.view(gc.size(0), gc.size(1), -1)[:, :num_words].transpose(1, 2)
        else:
            words_embs, sent_emb = b["words_embs"], b["sent_emb"]
        inv_world = 1.0 / self.world
        out = {}

        clabels_feat = form_clabels_feat(clabels_emb, rois[0], num_rois)
        # (1b) The discriminators' passes over the REAL images do not depend on the generator: they are issued now, each on
        # the side stream its update will run on, and execute while the generator's forward pass (one long chain on the main
        # stream, many of its launches too small to fill 256 CUs) is running.  Every network still sees real -> fake -> wrong
        # in that order on ONE stream (BatchNorm running statistics included); the python RNG is not touched here.
        side = self._d_side_streams()
        main = torch.cuda.current_stream() if side else None
        smap = self._stream_map_from_env() or self.d_stream_map or {}
        if isinstance(smap, (list, tuple)):             # (older callers: positional list in job order)
            smap = dict(zip(self._D_JOB_NAMES, smap))
        job_names = ["errPatD%d" % i for i in range(len(self.optimizersPatD))] + \
                    ["errShpD%d" % i for i in range(len(self.optimizersShpD))] + \
                    (["errObjSSD", "errObjLSD"] if self.use_obj else [])
        stream_of, rr = {}, 0
        for name in job_names:
            if name in smap:
                stream_of[name] = smap[name]
            else:
                stream_of[name], rr = rr, rr + 1
        pre = {}
        if side and self.hoist_real_passes:
            for s_ in side:
                s_.wait_stream(main)
            real_fns = [("errPatD%d" % i, opt, lambda i=i: patD_real(self.netsPatD[i], imgs[i]))
                        for i, opt in enumerate(self.optimizersPatD)] + \
                       [("errShpD%d" % i, opt, lambda i=i: shpD_real(self.netsShpD[i], imgs[i], hmaps[i]))
                        for i, opt in enumerate(self.optimizersShpD)]
            if self.use_obj:
                real_fns += [("errObjSSD", self.optimizerObjSSD,
                              lambda: objD_real(self.netObjSSD, imgs[-1], hmaps[-1], rois[0], num_rois)),
                             ("errObjLSD", self.optimizerObjLSD,
                              lambda: objD_real(self.netObjLSD, imgs[-1], hmaps[-1], fm_rois, num_rois))]
            for name, opt, fn in real_fns:
                with torch.cuda.stream(side[stream_of[name] % len(side)]):
                    opt.zero_grad()
                    pre[name] = fn()
        self._phase("start")
        # (2) generate fake images
        if noise is None:
            self.noise.normal_(0, 1)
            noise = self.noise
        glb_max_num_roi = int(_host(num_rois).max())
        fake_imgs, bt_c_codes, _, _, mu, logvar = self.netG(
            noise, sent_emb, words_embs, glove_words_embs, clabels_feat, mask, hmaps, rois,
            fm_rois, num_rois, b["bt_masks"], b["fm_bt_masks"], glb_max_num_roi)
        self._phase("g_forward")
        bt_c_codes = [c.detach() for c in bt_c_codes]
        self._damsm_pre = None
        if side and self.early_damsm and not want_logs:
            import miscc.losses as L
            st = self.__dict__.get("_damsm_stream")
            if st is None:
                st = self._damsm_stream = torch.cuda.Stream(device=self.device)
            st.wait_stream(main)
            with torch.cuda.stream(st):
                self._damsm_pre = (L.damsm_term(self._graphed("image_encoder"), fake_imgs[-1], words_embs, sent_emb,
                                                self.match_labels, b["cap_lens"], b["class_ids"]), st)

        # (3) the eight discriminator updates (reference trainer.py:398-443).  They are independent of
        # each other (own weights, the real batch, the detached fake images): the host issues them in the
        # reference's order, each on one of the side streams (`d_stream_map`, below).
        # The five users of the python RNG (permute_seg of the shape / object discriminators) get their class permutations
        # NOW, in the reference's order: the permutations are host work on the box tables and nothing else, so the order in
        # which the host issues the eight jobs below no longer has to follow the reference's (heaviest streams first).
        draws = {}
        if self.predraw_permutations:
            for i in range(len(self.optimizersShpD)):
                draws["errShpD%d" % i] = draw_class_permutations(hmaps[i], rois[i], num_rois)
            if self.use_obj:
                draws["errObjSSD"] = draw_class_permutations(hmaps[-1], rois[0], num_rois)
                draws["errObjLSD"] = draw_class_permutations(hmaps[-1], fm_rois, num_rois)
        jobs = []
        for i, opt in enumerate(self.optimizersPatD):
            jobs.append(("errPatD%d" % i, opt,
                         lambda real=None, i=i: patD_loss(self.netsPatD[i], imgs[i], fake_imgs[i], sent_emb, real=real)))
        for i, opt in enumerate(self.optimizersShpD):
            jobs.append(("errShpD%d" % i, opt,
                         lambda real=None, i=i: shpD_loss(self.netsShpD[i], imgs[i], fake_imgs[i], hmaps[i], rois[i], num_rois,
                                                          real=real, draw=draws.get("errShpD%d" % i))))
        # the reference updates an object discriminator only `if float(err) > 0`, i.e. when at least one
        # box of the wanted scale exists (BCE of a sigmoid is > 0 otherwise)
        obj_jobs = (("errObjSSD", self.netObjSSD, self.optimizerObjSSD, rois[0], False),
                    ("errObjLSD", self.netObjLSD, self.optimizerObjLSD, fm_rois, True)) if self.use_obj else ()
        for name, net, opt, r, large in obj_jobs:
            jobs.append((name, opt,
                         lambda real=None, net=net, r=r, large=large, name=name: objD_loss(net, imgs[-1], fake_imgs[-1], hmaps[-1],
                                                                                clabels_emb, bt_c_codes[-1], r, num_rois,
                                                                                is_large_scale=large, real=real,
                                                                                draw=draws.get(name))))
        # The eight discriminator updates read the same fake images and touch disjoint networks: they are spread
        # round-robin over `d_streams` HIP streams so that the many launches that do not fill 256 CUs on their own
        # (discriminator heads on 4x4 .. 16x16 maps, normalisation / combine kernels of small layers) overlap with
        # another discriminator's work.  The host enqueues them in the reference order (python RNG draws of
        # permute_seg included); autograd replays every backward node on the stream of its forward.
        order = ([n for n in os.environ["OBJGAN_D_ORDER"].split(",") if n] if os.environ.get("OBJGAN_D_ORDER")
                 else self.d_job_order)
        if order:
            by_name = dict((j[0], j) for j in jobs)
            jobs = [by_name[n] for n in resolve_d_job_order([j[0] for j in jobs], order, rng_free=bool(draws))]
        pending = []
        for s_ in side:
            s_.wait_stream(main)
        for j, (name, opt, loss_fn) in enumerate(jobs):
            ctx = torch.cuda.stream(side[stream_of[name] % len(side)]) if side else _NullCtx()
            with ctx:
                if name not in pre:
                    opt.zero_grad()
                err = loss_fn(pre.get(name))
                active = torch.is_tensor(err)
                if active:
                    wg = self._wgrad_side_stream(stream_of[name] % len(side) + 1) if (side and self.async_wgrad_d) else None
                    if wg is not None:              # this update's weight gradients beside its data-gradient chain
                        with ops.wgrad_stream_scope(wg):
                            err.backward()
                        ops.wgrad_join()            # (this side stream waits: the all-reduce / Adam of the arena follow)
                    else:
                        err.backward()
                    opt.arena.grad[-1:].fill_(1.0)       # "this rank has a gradient" flag
                    out[name] = err.detach()
                pending.append((opt, self._reduce_async(opt), active))
        for s_ in side:
            main.wait_stream(s_)

        # discriminator Adam steps (after their reductions; they overlapped the later Ds).  Under data
        # parallelism a rank cannot know on the host whether ANOTHER rank had boxes of the wanted scale:
        # the flag slot was summed by the all-reduce and gates the update on the device (no host read).
        for opt, handle, active in pending:
            if handle is not None:
                self._wait(handle)                   # stream-side wait, the host runs on
                # grad_scale < 0: mean over the ranks that CONTRIBUTED a gradient (the all-reduced flag counts them;
                # all of them for the patch / shape discriminators, possibly fewer for the object discriminators)
                opt.step(grad_scale=-1.0, gated=True)
            elif active:
                opt.step(grad_scale=inv_world)

        if self.debug_after_d_updates is not None:      # (tests: e.g. put the oracle's updated discriminators in place)
            self.debug_after_d_updates(self)
        self._phase("d_updates")
        # (4) generator: maximise log(D(G(z))) + DAMSM + KL, discriminators frozen
        d_opts = self._d_optimizers()
        for opt in d_opts:
            opt.arena.set_requires_grad(False)
        self.optimizerG.zero_grad()
        bt_last = bt_c_codes[-1] if bt_c_codes else None
        errG_total, G_logs = G_loss(self.netsPatD, self.netsShpD, self.netObjSSD, self.netObjLSD,
                                    self._graphed("image_encoder"), fake_imgs, hmaps, words_embs, sent_emb,
                                    clabels_emb, bt_last, self.match_labels, b["cap_lens"],
                                    b["class_ids"], rois[0], fm_rois, num_rois, use_obj=self.use_obj) \
            if want_logs else _g_loss_quiet(self, fake_imgs, hmaps, words_embs, sent_emb, clabels_emb,
                                            bt_last, b, rois, fm_rois, num_rois)
        kl = KL_loss(mu, logvar)
        errG_total = errG_total + kl
        # data parallel: the generator's gradient arena is all-reduced in buckets AS its backward pass
        # completes them (stage 3 first), not in one piece after it
        g_arena, handles = self.optimizerG.arena, []
        if self.ddp:
            def reduce_range(s0, e0):
                if hasattr(ops, "wgrad_join"):
                    ops.wgrad_join()            # (the bucket's weight gradients may still be running on the side stream)
                self._comm_note(g_arena.grad[s0:e0])
                handles.append(dist.all_reduce(g_arena.grad[s0:e0], op=dist.ReduceOp.SUM, async_op=True))
            g_arena.arm(reduce_range)
        self._phase("g_loss_forward")
        wg_stream = self._wgrad_side_stream()
        if wg_stream is not None:
            # the generator's backward pass is the one phase on a single stream: its weight gradients leave for a side
            # stream and the data-gradient chain goes on (ops.wgrad_stream_scope); ArenaAdam.step / the all-reduce join
            with ops.wgrad_stream_scope(wg_stream):
                errG_total.backward()
        else:
            errG_total.backward()
        for opt in d_opts:
            opt.arena.set_requires_grad(True)
        if self.ddp:
            during = len(handles)
            for s0, e0 in g_arena.disarm():
                reduce_range(s0, e0)
            self.g_buckets = (during, len(handles) - during)     # issued inside backward / after it
            for h in handles:
                self._wait(h)
        self._phase("g_backward")
        self.optimizerG.step(grad_scale=inv_world)
        ops.ema_update_(self.avg_param_G, self.optimizerG.arena.flat, 0.999)
        out["errG"] = errG_total.detach()
        out["kl"] = kl.detach()
        out["fake_imgs"] = [f.detach() for f in fake_imgs]
        if want_logs:
            out["G_logs"] = G_logs

        self._phase("g_update")
        # (5) Inception-score monitor on a side stream (no data dependence on the update)
        if self.inception_model is not None:
            img = out["fake_imgs"][-1]
            self.is_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.is_stream), torch.no_grad():
                pred = self._graphed("inception_model")(img)
                if isinstance(pred, tuple):              # (a replayed graph hands out its static output buffer: the
                    pred = pred[0].clone()               #  caller keeps predictions across steps, write_scores)
            # the image was allocated on the main stream and is read on the side stream; the prediction
            # the other way round: tell the caching allocator, and remember where to wait before reading
            img.record_stream(self.is_stream)
            pred.record_stream(torch.cuda.current_stream())
            out["is_pred"] = pred
        self.gen_iterations += 1
        return out

    # ---- sampling (reference trainer.py:275-283, evaluator.py:330-340) -----------------------------
    @torch.no_grad()
    def sample(self, batch, noise, words_embs, sent_emb, glove_words_embs, mask, use_ema=True):
        """Generator inference in eval mode (BatchNorm on its running statistics), by default with the
        EMA weights swapped into the arena for the duration of the call -- what the reference does with
        copy_G_params / load_params around save_img_results.  The swap goes through the arena, so its
        epoch is bumped on the way in and out: packed filter banks are keyed on it.
        -> (fake_imgs, attention maps, bottom-up attention maps)"""
        b = batch
        arena = self.optimizerG.arena
        backup = None
        if use_ema:
            backup = arena.flat.clone()
            arena.flat.copy_(self.avg_param_G)
            arena.epoch[0] += 1
        was_training = self.netG.training
        self.netG.eval()
        try:
            clabels_feat = form_clabels_feat(b["clabels_emb"], b["rois"][0], b["num_rois"])
            glb = int(_host(b["num_rois"]).max())
            fake_imgs, _, att, bt_att, _, _ = self.netG(
                noise, sent_emb, words_embs, glove_words_embs, clabels_feat, mask, b["hmaps"], b["rois"],
                b["fm_rois"], b["num_rois"], b["bt_masks"], b["fm_bt_masks"], glb)
        finally:
            self.netG.train(was_training)
            if backup is not None:
                arena.flat.copy_(backup)
                arena.epoch[0] += 1
        return fake_imgs, att, bt_att

    # ---- checkpoints (reference trainer.py:251-273) ---------------------------------------------------
    def save_model(self, netG, avg_param_G, netsPatD, netsShpD, netObjSSD, netObjLSD, epoch):
        if self.rank != 0:
            return
        arena = self.optimizerG.arena
        backup = arena.flat.clone()
        arena.flat.copy_(avg_param_G)                 # G is saved with the EMA weights swapped in
        torch.save(netG.state_dict(), '%s/netG_epoch_%d.pth' % (self.model_dir, epoch))
        arena.flat.copy_(backup)
        for i, d in enumerate(netsPatD):
            torch.save(d.state_dict(), '%s/netPatD%d.pth' % (self.model_dir, i))
        for i, d in enumerate(netsShpD):
            torch.save(d.state_dict(), '%s/netShpD%d.pth' % (self.model_dir, i))
        torch.save(netObjSSD.state_dict(), '%s/netObjSSD.pth' % self.model_dir)
        torch.save(netObjLSD.state_dict(), '%s/netObjLSD.pth' % self.model_dir)

    def join_monitor(self):
        """Make the current stream wait for the Inception-score monitor's side stream: call before
        reading any `is_pred` (write_scores does)."""
        if getattr(self, "is_stream", None) is not None:
            torch.cuda.current_stream().wait_stream(self.is_stream)

    def write_scores(self, predictions, epoch):
        """Per-epoch Inception score of the per-step monitor predictions -> Score/scores_<epoch>.txt
        (reference trainer.py:495-506)."""
        join = getattr(self, "join_monitor", None)
        if join is not None:
            join()
        preds = np.concatenate([p.detach().cpu().numpy() if torch.is_tensor(p) else np.asarray(p)
                                for p in predictions], 0)
        splits = min(10, self.batch_size)
        mean, std = compute_inception_score(preds, splits)
        mean_conf, std_conf = negative_log_posterior_probability(preds, splits)
        with open('%s/scores_%d.txt' % (self.score_dir, epoch), 'w') as fp:
            fp.write('mean, std, mean_conf, std_conf \n')
            fp.write('%f, %f, %f, %f' % (mean, std, mean_conf, std_conf))
        print('inception_score: %f, %f, %f, %f' % (mean, std, mean_conf, std_conf))
        return mean, std, mean_conf, std_conf

    # ---- epoch loop -------------------------------------------------------------------------------------
    def train(self):
        self.setup()
        for epoch in range(self.start_epoch, self.max_epoch):
            start_t = time.time()
            sampler = getattr(self.data_loader, "sampler", None)
            if hasattr(sampler, "set_epoch"):       # reshuffle the per-rank shards every epoch
                sampler.set_epoch(epoch)
            predictions = []
            for step, batch in enumerate(self.data_loader):
                if not isinstance(batch, dict):       # the reference loader's collated 12-tuple
                    from trainDataset import prepare_data, batch_dict
                    batch = batch_dict(prepare_data(batch, self.device, self.num_classes), self.clabels_emb)
                log_now = (self.gen_iterations + 1) % self.print_interval == 0
                out = self.train_step(batch, want_logs=log_now)
                if "is_pred" in out:
                    predictions.append(out["is_pred"])
                if log_now and self.rank == 0:
                    msg = ' '.join('%s: %.2f' % (k, float(v)) for k, v in out.items()
                                   if torch.is_tensor(v) and v.dim() == 0)
                    print('[%d/%d][%d] %s %s' % (epoch, self.max_epoch, self.gen_iterations, msg,
                                                 out.get("G_logs", "")))
            if self.rank == 0:
                print('[%d/%d] time: %.2fs' % (epoch, self.max_epoch, time.time() - start_t))
                if predictions:
                    self.write_scores(predictions, epoch)
            if epoch % self.snapshot_interval == 0:
                self.save_model(self.netG, self.avg_param_G, self.netsPatD, self.netsShpD,
                                self.netObjSSD, self.netObjLSD, epoch)
        self.save_model(self.netG, self.avg_param_G, self.netsPatD, self.netsShpD, self.netObjSSD,
                        self.netObjLSD, self.max_epoch)


def _g_loss_quiet(tr, fake_imgs, hmaps, words_embs, sent_emb, clabels_emb, bt_c, b, rois, fm_rois,
                  num_rois):
    """G_loss without building the log string (no .item() host syncs on non-print steps)."""
    import miscc.losses as L
    total, _ = L.G_loss(tr.netsPatD, tr.netsShpD, tr.netObjSSD, tr.netObjLSD, tr._graphed("image_encoder"),
                        fake_imgs, hmaps, words_embs, sent_emb, clabels_emb, bt_c, tr.match_labels,
                        b["cap_lens"], b["class_ids"], rois[0], fm_rois, num_rois, quiet=True,
                        use_obj=tr.use_obj, streams=tr._d_side_streams() or None,
                        damsm_pre=tr.__dict__.pop("_damsm_pre", None))
    return total, ''
