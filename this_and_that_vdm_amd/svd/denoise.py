"""The fused denoise loop: the body of StableVideoDiffusion(ControlNet)Pipeline.__call__'s step loop
(svd/pipeline_stable_video_diffusion_controlnet.py:624-720, VL twin svd/pipeline_stable_video_diffusion.py:528-562)
as ONE hipGraph replayed per step.

Per step, on device:   prep (CFG duplicate, x/sqrt(sigma^2+1), channel concat, NCHW->tokens)  ->  time/FiLM rows
  ->  UNet encoder+mid  ->  GestureNet encoder+mid, zero-convs with the UNet skips added in the epilogue
  ->  UNet decoder  ->  per-frame CFG + v-prediction Euler update of the fp32 latents (in place).
Hoisted out of the loop (step-invariant; the reference recomputes them every step):
context K/V of all 46 cross-attention layers, frame-position embeddings, and the gesture-map latents (the
pipeline VAE-encodes them once instead of 25 times, reference :652).  The time embedding + FiLM rows of all ResBlocks depend
on the schedule only: begin() evaluates them for all steps at once (DenoiserBase.film_table, round 4) and step i copies its
row block into the static buffer the captured epilogues read (TT_FILM_TABLE=0: per step inside the graph, as before).
Step-dependent scalars (sigma_i, sigma_{i+1}, t_i) live in a 3-float device buffer that is refreshed by one
tiny copy before each replay, so a single captured graph serves all steps.  No host sync inside the loop.
Control-guidance windows (reference :611-617,639-645): a step whose ``controlnet_keep`` is 0 multiplies every residual
by 0, i.e. it is exactly the UNet-only step -- such steps replay a second graph that does not launch GestureNet at all."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

import os

from .. import ops
from .layers import Geom

FILM_TABLE = os.environ.get("TT_FILM_TABLE", "1") != "0"


class DenoiseLoop:
    def __init__(self, unet, controlnet=None, use_graph: bool = True, split_cfg: bool = False):
        self.unet, self.controlnet, self.use_graph = unet, controlnet, use_graph
        # run the uncond / cond halves as separate concurrent branches too (measured SLOWER on MI355X at 256x448:
        # 47.7 vs 44.4 ms/step -- half-size GEMMs lose more than the overlap wins; kept as an option and tested)
        self.split_cfg = split_cfg
        self.overlap_branches = True        # False: same launches, one stream (used when timing kernels one by one)
        self._streams = {}
        self._graph = None
        self._graph_off = None              # the UNet-only step for controlnet_keep[i] == 0
        self._key = None
        self._static = {}

    # ---- request set-up (everything step-invariant)
    def begin(self, latents: torch.Tensor, image_latents: torch.Tensor, encoder_hidden_states: torch.Tensor,
              added_time_ids: torch.Tensor, guidance_scale: Optional[torch.Tensor], sigmas: torch.Tensor,
              timesteps: torch.Tensor, controlnet_cond: Optional[torch.Tensor] = None, conditioning_scale: float = 1.0,
              controlnet_keep: Optional[Sequence[float]] = None, image_guidance_scale: Optional[float] = None,
              guess_mode: bool = False):
        """latents [1,F,4,h,w] (already scaled by init_noise_sigma); image_latents [B,F,4,h,w]; encoder_hidden_states
        [B,S,D]; added_time_ids [B,3]; guidance_scale [1,F,1,1,1] or None (no CFG: B == 1); sigmas [steps+1],
        timesteps [steps]; controlnet_cond [F,4,h,w] gesture latents (same for both CFG halves, reference :660); controlnet_keep: one 0.0/1.0 per step
        (reference :611-617), None = keep everywhere.  B == 3 is the use_instructpix2pix batch (first-frame, cond, uncond; reference
        :182-184,208-210,698-702) and needs image_guidance_scale.  guess_mode: the 13 residual scales become
        logspace(-1, 0, 13) * conditioning_scale (temporal_controlnet.py:626-630); only without CFG -- the reference's
        guess-mode + CFG branch (:676-681) concatenates zeros onto an already CFG-sized residual batch and cannot run."""
        dev = self.unet.device
        self.unet.prepare()
        b = image_latents.shape[0]
        _, f, _, h, w = latents.shape
        if b not in (1, 2, 3):
            raise ValueError("CFG batch must be 1 (no CFG), 2 (uncond, cond) or 3 (use_instructpix2pix)")
        if b == 3 and (image_guidance_scale is None or guidance_scale is None):
            raise ValueError("a CFG batch of 3 (use_instructpix2pix) needs guidance_scale and image_guidance_scale")
        self.image_guidance_scale = float(image_guidance_scale) if b == 3 else None
        dtype = self.unet._run_dtype()
        if self.controlnet is not None:
            self.controlnet.prepare()
        # a captured graph holds raw pointers into the models' packed weights: the pack generation of both models is part
        # of the key, so load_state_dict / .to() / in-place updates between requests drop the stale graphs
        packs = self._pack_state()
        key = (b, f, h, w, dtype, self.controlnet is not None, tuple(encoder_hidden_states.shape), len(timesteps),
               guidance_scale is not None, self.image_guidance_scale, packs,   # the image scale is baked into the graph
               ops.f32_split())                                                # ... and so is the TT_F32 product mode (kernel variants)
        if key != self._key:                    # new shapes or new weights: new static buffers, new graph
            self._graph, self._graph_off, self._key, self._static = None, None, key, {}
        self._packs = packs                     # what the FiLM table, the context projections and the graphs below were built from
        self.geom, self.dtype = Geom(b, f, h, w), dtype
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        sig = f32(sigmas)
        # static buffers: a captured graph keeps raw pointers, so later requests are COPIED into the same storage
        self.latents = self._static_set("latents", f32(latents).reshape(f, 4, h, w))
        self.image_latents = self._static_set("image_latents", f32(image_latents))
        self.added_time_ids = self._static_set("added_time_ids", f32(added_time_ids))
        self.guidance = self._static_set("guidance", f32(guidance_scale).reshape(-1)) if guidance_scale is not None else None
        self.table = self._static_set("table", torch.stack([sig[:-1], sig[1:], f32(timesteps)], 1))       # [steps, 3]
        self.cur = self._static_set("cur", torch.zeros(3, dtype=torch.float32, device=dev))           # sigma, sigma_next, t
        self.num_steps = self.table.shape[0]
        self.film_tab_u = self.film_tab_c = self.film_cur_u = self.film_cur_c = None
        if FILM_TABLE:
            self.film_tab_u = self._static_set("film_tab_u", self.unet.film_table(self.table[:, 2], self.added_time_ids, b))
            self.film_cur_u = self._static_set("film_cur_u", self.film_tab_u[0])
        ehs = encoder_hidden_states.to(dev)
        k, vt, s, sp, zmask = self.unet.project_context(ehs)
        if self._static.setdefault("zero_ctx_mask", zmask) != zmask:
            self._graph = self._graph_off = None        # which rows skip cross-attention is part of the launch structure
            self._static["zero_ctx_mask"] = zmask
        self.ctx_unet = (self._static_set("k_unet", k), self._static_set("vt_unet", vt), s, sp, zmask)
        self.cond = None
        if self.controlnet is not None:
            if guess_mode and b > 1:
                raise NotImplementedError("guess_mode with CFG (the reference's branch :676-681 cannot run either)")
            if controlnet_cond is None:
                raise ValueError("controlnet_cond (VAE-encoded gesture latents) is required with a ControlNet")
            self.controlnet.prepare()
            if self.controlnet._run_dtype() != self.dtype:
                raise RuntimeError("UNet and ControlNet must run in the same 16-bit dtype inside the fused loop")
            self.cond = self._static_set("cond", f32(controlnet_cond).reshape(f, 4, h, w))
            if FILM_TABLE:
                self.film_tab_c = self._static_set("film_tab_c", self.controlnet.film_table(self.table[:, 2], self.added_time_ids, b))
                self.film_cur_c = self._static_set("film_cur_c", self.film_tab_c[0])
            k, vt, s, sp, zmask_cn = self.controlnet.project_context(ehs)
            if self._static.setdefault("zero_ctx_mask_cn", zmask_cn) != zmask_cn:
                self._graph = self._graph_off = None
                self._static["zero_ctx_mask_cn"] = zmask_cn
            self.ctx_cn = (self._static_set("k_cn", k), self._static_set("vt_cn", vt), s, sp, zmask_cn)
            self.cn_scales = self.controlnet._scales(float(conditioning_scale), bool(guess_mode), len(self.controlnet.controlnet_down_blocks))
            if self._static.setdefault("cn_scales", self.cn_scales) != self.cn_scales:
                self._graph = None              # the scale is a launch argument baked into the graph
                self._static["cn_scales"] = self.cn_scales
        self.keep = [1.0] * self.num_steps if controlnet_keep is None else [float(v) for v in controlnet_keep]
        if len(self.keep) != self.num_steps or any(v not in (0.0, 1.0) for v in self.keep):
            raise ValueError("controlnet_keep needs one 0.0/1.0 entry per step")
        self.step_index = 0
        return self

    def _pack_state(self, repack: bool = True):
        """(identity, pack generation, fp8 switch) of both models: prepare() bumps the generation on every repack.
        repack=True (begin): prepare() first, so parameters that changed since the last pack are re-packed now.
        repack=False (step): nothing is packed; a model whose parameters no longer match its pack (in-place update, .to(),
        load_state_dict, invalidate_packs) reports generation -1, which never equals what begin() recorded."""
        st = ()
        for m in (self.unet, self.controlnet):
            if m is None:
                continue
            if repack:
                m.prepare()
            fresh = repack or getattr(m, "_packed_key", None) == m._pack_key()
            st += (id(m), m._pack_gen if fresh else -1, bool(m.attention_fp8))
        return st

    def _static_set(self, name: str, value: torch.Tensor) -> torch.Tensor:
        cur = self._static.get(name)
        if cur is not None and cur.shape == value.shape and cur.dtype == value.dtype:
            cur.copy_(value)
            return cur
        self._static[name] = value.clone()          # never alias the caller's tensor: latents are updated in place
        if cur is not None:
            self._graph = self._graph_off = None
        return self._static[name]

    # ---- one step's launches (captured once)
    def _launch_step(self, use_cn: bool = True):
        """One step as concurrent branches (fork/join with events; captured as parallel hipGraph branches).
        GestureNet's encoder+mid is independent of the UNet's until the zero-convs, so the two run side by side: the
        latency-bound small kernels and the tails of one branch fill the idle CUs of the other (49.8 -> 44.4 ms/step).
        With ``split_cfg`` the uncond / cond halves (which never interact inside the networks) become branches too."""
        g = self.geom
        cn = self.controlnet if use_cn else None
        from . import layers as _layers
        _layers._Side.origin = torch.cuda.current_stream().cuda_stream     # the only stream that may fork side launches under capture
        cpad = cn._cin_pad if cn is not None else self.unet._cin_pad
        x_tok = ops.prep_model_input(self.latents, self.image_latents, self.cond if use_cn else None, self.cur, 0, g.batch, g.frames, g.h, g.w,
                                     cpad, self.dtype)
        t = self.cur[2:3]
        x_unet = x_tok if cpad == self.unet._cin_pad else x_tok[:, :self.unet._cin_pad]
        if self.film_cur_u is not None:         # this step's rows of the per-request FiLM table (copied in by step())
            ctx_u = self.unet._step_context(None, self.ctx_unet, film=self.film_cur_u)
        else:
            ctx_u = self.unet._step_context(self.unet._embed(t, self.added_time_ids, g.batch, x_tok.device), self.ctx_unet)
        ctx_c = None
        if cn is not None:
            if self.film_cur_c is not None:
                ctx_c = cn._step_context(None, self.ctx_cn, film=self.film_cur_c)
            else:
                ctx_c = cn._step_context(cn._embed(t, self.added_time_ids, g.batch, x_tok.device), self.ctx_cn)
        eps = torch.empty((g.m, self.unet.conv_out.out_channels), dtype=torch.float32, device=x_tok.device)
        halves = [Geom(1, g.frames, g.h, g.w, b, g.batch) for b in range(g.batch)] if self.split_cfg and g.batch > 1 else [g]
        rows = halves[0].m
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        tails = []
        for i, gh in enumerate(halves):
            su = main if i == 0 else self._stream(f"u{i}")
            lo, hi_ = i * rows, (i + 1) * rows
            join = None
            if cn is not None:
                sc = self._stream(f"c{i}")
                sc.wait_event(fork)
                with torch.cuda.stream(sc):
                    cn_skips, cn_mid, _ = cn.encode_tokens(x_tok[lo:hi_], gh, ctx_c)
                    join = torch.cuda.Event()
                    join.record(sc)
            if su is not main:
                su.wait_event(fork)
            with torch.cuda.stream(su):
                x, gm, skips = self.unet.encode_tokens(x_unet[lo:hi_], gh, ctx_u)
                if cn is not None:
                    su.wait_event(join)
                    skips, (x, _) = cn.zero_convs(cn_skips, cn_mid, gm, self.cn_scales, add_to=([s for s, _ in skips], x))
                self.unet.decode_tokens(x, gm, skips, ctx_u, eps_out=eps[lo:hi_])
                if su is not main:
                    done = torch.cuda.Event()
                    done.record(su)
                    tails.append(done)
        for ev in tails:
            main.wait_event(ev)
        ops.cfg_euler_step(eps, self.latents, self.guidance, self.cur, 0, g.batch, g.frames, g.h, g.w,
                           self.image_guidance_scale)

    def _stream(self, name):
        if not self.overlap_branches:
            return torch.cuda.current_stream()
        st = self._streams.get(name)
        if st is None:
            st = self._streams[name] = torch.cuda.Stream()
        return st

    def step(self):
        """Advance the latents by one Euler step (asynchronous; call torch.cuda.synchronize() to wait)."""
        if self.step_index >= self.num_steps:
            raise RuntimeError("denoise loop already finished; call begin() for a new request")
        # begin() evaluated the FiLM rows of ALL steps and the context K/V from the weights of that moment, and the captured graph
        # holds pointers into that pack: a weight reload / repack between begin() and step() must not be served from them
        if self._pack_state(repack=False) != self._packs:
            raise RuntimeError("model weights were re-packed (load_state_dict / .to() / in-place update) after DenoiseLoop.begin(): "
                               "call begin() again -- the per-request FiLM table and context projections belong to the old weights")
        self.cur.copy_(self.table[self.step_index])
        use_cn = self.controlnet is not None and self.keep[self.step_index] != 0.0
        if self.film_cur_u is not None:
            self.film_cur_u.copy_(self.film_tab_u[self.step_index])
            if use_cn:
                self.film_cur_c.copy_(self.film_tab_c[self.step_index])
        if not self.use_graph:
            self._launch_step(use_cn)
        elif use_cn or self.controlnet is None:
            if self._graph is None:
                self._graph = self._capture(self.controlnet is not None)
            self._graph.replay()
        else:
            if self._graph_off is None:
                self._graph_off = self._capture(False)
            self._graph_off.replay()
        self.step_index += 1

    def _capture(self, use_cn: bool):
        # warm-up on a side stream (allocator + lazily built caches), restoring the latents afterwards
        keep = self.latents.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._launch_step(use_cn)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._launch_step(use_cn)
        self.latents.copy_(keep)
        return graph

    def run(self, steps: Optional[int] = None) -> torch.Tensor:
        n = self.num_steps - self.step_index if steps is None else steps
        for _ in range(n):
            self.step()
        return self.result()

    def result(self) -> torch.Tensor:
        g = self.geom
        return self.latents.view(1, g.frames, 4, g.h, g.w)
