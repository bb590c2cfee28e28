"""DDIM sampler on MI355X: drop-in for `audioldm2.latent_diffusion.models.ddim.DDIMSampler`
(models/ddim.py:15-355).  Same constructor / `sample()` / `ddim_sampling()` / `p_sample_ddim()`
signatures and return values; constructed by name in `LatentDiffusion.sample_log` (ddpm.py:1437), so a
reference user rebinds `ddpm.DDIMSampler = audioldm2_amd.ddim.DDIMSampler`.

Differences that matter on MI355X:
  * classifier-free guidance runs as ONE UNet pass over [uncond ; cond] (2B samples) when the model
    exposes `apply_model_cfg` (the reference runs two sequential passes, ddim.py:293-296);
  * CFG combine + x0 prediction + x_{t-1} update is one fused kernel (ops.ddim_step);
  * the per-step launch sequence (~1000 kernels) is captured once into a HIP graph and replayed;
    timestep, DDIM coefficients and noise are device-side inputs of the graph;
  * RNG contract: x_T and the per-step noise are drawn from the HOST CPU
    generator in the reference's order and shapes (ddim.py:191,351 via util.py:289-294), uploaded
    once — results match the CPU reference on the same seed.
"""
from __future__ import annotations

import gc
import os

import numpy as np
import torch

from . import ops


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """util.py:55-75"""
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ddim_timesteps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        ddim_timesteps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * 0.8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    return ddim_timesteps + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """util.py:78-95 — keeps the reference's mixed float32-tensor / float64-ndarray arithmetic:
    (1 - alphas) is a float32 tensor op, the rest float64."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return torch.as_tensor(np.asarray(sigmas, dtype=np.float64)), alphas, alphas_prev


class GraphStepper:
    """Runs a fixed launch sequence repeatedly: eagerly the first time (packs weights, warms the
    allocator, grows the split-K workspace), captured into a HIP graph the second time, replayed after.
    The callable must read its per-step inputs from static device buffers."""

    def __init__(self, fn, use_graph=True):
        self.fn, self.use_graph, self.calls, self.graph = fn, use_graph, 0, None

    def __call__(self):
        if self.use_graph and self.calls >= 1:
            if self.graph is None:
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                # No cyclic garbage collection while the stream is capturing: a collection that happens to run inside the
                # capture can finalise objects of EARLIER jobs (an evicted step graph, a noise feed's side stream and pinned
                # buffers) whose destructors call HIP functions that are illegal during capture — the C++ destructor then
                # throws and the process aborts ("Fatal Python error: Aborted ... Garbage-collecting" a few tests into a
                # long session, at a different place every time).  Collect first, then hold the collector off.
                gc.collect()
                gc_was_on = gc.isenabled()
                gc.disable()
                try:
                    # thread_local: other threads (e.g. the RCCL watchdog of a multi-GPU run) may keep
                    # calling HIP while this thread captures
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self.fn()
                    self.graph = g
                except RuntimeError as e:  # capture refused: keep sampling eagerly, say so once
                    import sys
                    print(f"[audioldm2_amd] HIP graph capture failed ({e}); continuing without a graph",
                          file=sys.stderr, flush=True)
                    self.use_graph = False
                    torch.cuda.synchronize()
                    if gc_was_on:
                        gc.enable()
                    self.fn()
                    self.calls += 1
                    return
                finally:
                    if gc_was_on:
                        gc.enable()
            self.graph.replay()
        else:
            self.fn()
        self.calls += 1


def drop_graph_entries(cache: dict) -> None:
    """Evict cached step graphs NOW, by reference counting: an entry is a reference cycle (entry -> stepper -> step closure ->
    entry), so simply forgetting it leaves the HIP graph, its private memory pool and its static buffers to the cyclic
    collector — which may then run in the middle of the next capture (see GraphStepper).  Breaking the cycle here frees
    them at a point where no stream is capturing."""
    for ent in list(cache.values()):
        rs = ent.get("run_step")
        if rs is not None:
            rs.fn = None
            rs.graph = None
        ent.clear()
    cache.clear()


class _NoiseFeed:
    """Streams the per-step host noise to the GPU in chunks.  A DRAWER THREAD consumes the CPU generator (the reference's RNG
    stream, strictly in its order: only this thread draws while a sampling run is in flight) into pinned staging rows and
    issues the PCIe uploads on a side stream, running up to `len(pin)` chunks ahead of the GPU; the launching thread only
    makes its stream wait for a chunk's upload event (`wait(i)`).  Round 2 drew on the launching thread between graph replays:
    with prompt sharding every rank draws the GLOBAL batch (7 ms per step at 8 ranks x 8 prompts against a 16 ms GPU step,
    profiles/r02_host_noise_cost.txt) — one kernel speed-up away from bounding the job.  `torch.randn` releases the GIL.
    Device buffers hold all steps (`noise_buf` / `qnoise_buf`: write into these — the static inputs of a cached step graph —
    instead of allocating); `close()` joins the thread: after it the generator is where the reference leaves it."""

    def __init__(self, draw, steps, shape, with_mask, temperature, dev, chunk=8, mask_first=True, noise_buf=None,
                 qnoise_buf=None, threaded=None):
        import threading
        self.draw, self.steps, self.with_mask, self.temperature = draw, steps, with_mask, temperature
        self.mask_first = mask_first  # DDIM draws the q_sample noise before the step noise, DDPM after
        # chunk boundaries: the first chunk is ONE step so the GPU starts right away, then `chunk` steps
        self.bounds = [0, min(1, steps)]
        while self.bounds[-1] < steps:
            self.bounds.append(min(steps, self.bounds[-1] + chunk))
        self.chunk = chunk
        self.noise = noise_buf if noise_buf is not None else torch.empty((steps,) + tuple(shape), device=dev,
                                                                         dtype=torch.float32)
        assert self.noise.shape == (steps,) + tuple(shape)
        self.qnoise = (qnoise_buf if qnoise_buf is not None else torch.empty_like(self.noise)) if with_mask else None
        nbuf = 3
        self.pin = [torch.empty((chunk,) + tuple(shape)).pin_memory() for _ in range(nbuf)]
        self.qpin = [torch.empty((chunk,) + tuple(shape)).pin_memory() for _ in range(nbuf)] if with_mask else None
        self.stream = torch.cuda.Stream(device=dev)
        # the device buffers may have been handed out while kernels already queued on the current stream (the previous run's
        # noise copies / step graph) still read them: the upload stream must not start writing before the current stream has
        # reached this point
        self.stream.wait_stream(torch.cuda.current_stream())
        self.noise.record_stream(self.stream)
        if self.qnoise is not None:
            self.qnoise.record_stream(self.stream)
        self.events = {}
        self.produced = 0  # chunks drawn so far
        self.dev_index = torch.cuda.current_device()   # the drawer thread must issue its uploads on THIS device
        self.error = None
        self.cv = threading.Condition()
        self.threaded = (os.environ.get("ALDM_NOISE_THREAD", "1") != "0") if threaded is None else threaded
        self.thread = None
        if self.threaded:
            self.thread = threading.Thread(target=self._run, name="aldm-noise-drawer", daemon=True)
            self.thread.start()

    def _draw_into(self, dst):
        """One reference-ordered draw into a (pinned) staging row."""
        d = self.draw
        if getattr(d, "into", None) is not None:
            d.into(dst)
        else:
            dst.numpy()[...] = d().numpy()   # plain memcpy, no thread pool

    def _produce(self, c):
        lo, hi = self.bounds[c], self.bounds[c + 1]
        slot = c % len(self.pin)
        prev = self.events.get(c - len(self.pin))
        if prev is not None:
            prev.synchronize()  # the upload that last used this pinned slot is done
        # Host side of the RNG contract.  The draws go STRAIGHT into the pinned staging rows (`torch.randn(out=...)`: same
        # generator stream, no intermediate tensor): a 1 MB `Tensor.copy_` through torch's intra-op thread pool costs
        # ~20 ms on a many-core host (measured: 19 ms with 8 threads on 8 busy vCPUs vs 0.02 ms with 4).
        for s in range(lo, hi):  # reference order per step
            if self.with_mask and self.mask_first:
                self._draw_into(self.qpin[slot][s - lo])
            self._draw_into(self.pin[slot][s - lo])
            if self.temperature != 1.0:
                self.pin[slot][s - lo].mul_(self.temperature)
            if self.with_mask and not self.mask_first:
                self._draw_into(self.qpin[slot][s - lo])
        with torch.cuda.stream(self.stream):
            self.noise[lo:hi].copy_(self.pin[slot][:hi - lo], non_blocking=True)
            if self.with_mask:
                self.qnoise[lo:hi].copy_(self.qpin[slot][:hi - lo], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return ev

    def _run(self):   # drawer thread
        try:
            torch.cuda.set_device(self.dev_index)
            for c in range(len(self.bounds) - 1):
                ev = self._produce(c)
                with self.cv:
                    self.events[c] = ev
                    self.produced = c + 1
                    self.cv.notify_all()
        except BaseException as e:  # surfaced by the launching thread's next wait()
            with self.cv:
                self.error = e
                self.cv.notify_all()

    def produce_next(self):
        """Unthreaded mode (ALDM_NOISE_THREAD=0 / tests): draw + upload the next chunk on the calling thread."""
        if self.threaded:
            return
        c = self.produced
        if c + 1 >= len(self.bounds):
            return
        self.events[c] = self._produce(c)
        self.produced = c + 1

    def wait(self, i):
        """Make the current stream wait for step i's noise; returns True when i opens a chunk."""
        c = 0 if i == 0 else 1 + (i - 1) // self.chunk
        if self.threaded:
            with self.cv:
                while self.produced <= c and self.error is None:
                    self.cv.wait(timeout=60.0)
                if self.error is not None:
                    raise RuntimeError(f"noise drawer thread failed: {self.error!r}")
        else:
            while self.produced <= c:
                self.produce_next()
        first = i == self.bounds[c]
        if first:
            torch.cuda.current_stream().wait_event(self.events[c])
        return first

    def close(self):
        """Join the drawer: every draw of the run has been consumed from the generator (callers draw again afterwards)."""
        if self.thread is not None:
            self.thread.join()
            self.thread = None
            if self.error is not None:
                raise RuntimeError(f"noise drawer thread failed: {self.error!r}")


def host_drawer(shape, noise_shard=None):
    """draw() -> one `torch.randn(shape)` from the host default generator, in the reference's order; draw.into(dst) writes
    it into `dst`.  noise_shard = (global_batch, row_offset): a prompt-sharded run draws the GLOBAL batch like the
    single-process reference and keeps rows [offset, offset + shape[0]) (dist.py); with several candidates per prompt
    the second element is the index tensor of the rank's rows (dist.candidate_rows) instead of an offset."""
    shape = tuple(shape)
    if noise_shard is None:
        def draw():
            return torch.randn(shape)

        def into(dst):
            torch.randn(shape, out=dst)
    else:
        gB, off = noise_shard
        gshape = (gB,) + shape[1:]
        if torch.is_tensor(off):
            assert off.numel() == shape[0]
            pick = lambda t: t.index_select(0, off)
        else:
            pick = lambda t: t[off:off + shape[0]]

        def draw():
            return pick(torch.randn(gshape)).contiguous()

        def into(dst):
            dst.numpy()[...] = pick(torch.randn(gshape)).numpy()
    draw.into = into
    return draw


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", device=torch.device("cuda"), **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.device = device
        self.use_graph = os.environ.get("ALDM_NO_GRAPH", "0") != "1"
        # (global_batch, row_offset) when this process samples one shard of a larger batch (dist.py)
        self.noise_shard = getattr(model, "noise_shard", None)

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0.0, verbose=True):
        """ddim.py:33-91 (host side, float tables only)."""
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose)
        alphas_cumprod = self.model.alphas_cumprod.detach().float().cpu()
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        self.alphas_cumprod = alphas_cumprod
        self.sqrt_alphas_cumprod = torch.sqrt(alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = torch.sqrt(1.0 - alphas_cumprod)
        sig, a, a_prev = make_ddim_sampling_parameters(alphas_cumprod, self.ddim_timesteps, ddim_eta, verbose)
        self.ddim_sigmas = sig
        self.ddim_alphas = a
        self.ddim_alphas_prev = a_prev
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1.0 - a.numpy())
        # per-index coefficient rows with the reference's roundings (ddim.py:330-353): a_t, a_prev,
        # sigma_t, sqrt(1-a_t) become fp32 via torch.full; the other square roots are fp32 tensor ops.
        rows = []
        for i in range(len(self.ddim_timesteps)):
            a_t = torch.full((1,), float(a[i]))
            ap = torch.full((1,), float(a_prev[i]))
            sg = torch.full((1,), float(sig[i]))
            som = torch.full((1,), float(self.ddim_sqrt_one_minus_alphas[i]))
            rows.append(torch.cat([som, a_t.sqrt(), (1.0 - ap - sg ** 2).sqrt(), ap.sqrt(), sg]))
        self.ddim_coef = torch.stack(rows)  # [S, 5] fp32 (host)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0.0, mask=None, x0=None, temperature=1.0,
               noise_dropout=0.0, score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None,
               log_every_t=100, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               dynamic_threshold=None, ucg_schedule=None, **kwargs):
        """ddim.py:94-163"""
        if quantize_x0 or score_corrector is not None or dynamic_threshold is not None \
                or noise_dropout != 0.0 or ucg_schedule is not None:
            raise NotImplementedError("DDIMSampler(HIP): option not used by the AudioLDM2 pipeline")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  mask=mask, x0=x0, temperature=temperature, x_T=x_T,
                                  log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning)

    # ------------------------------------------------------------------------------------------
    def _drawer(self, shape):
        """One reference-ordered Gaussian draw from the HOST default generator (RNG contract R).  The returned callable
        also has `.into(dst)`: the same draw written into an existing host tensor."""
        return host_drawer(shape, self.noise_shard)

    def _draw_noise(self, shape, steps, x_T, with_mask):
        """Replays the reference's host RNG order: x_T, then per step [q_sample noise (inpainting
        only, ddim.py:228 / ddpm.py:430-436)], step noise (ddim.py:351).  (Eager form, used by tests;
        ddim_sampling streams the same sequence through _NoiseFeed.)"""
        draw = self._drawer(shape)
        img = draw() if x_T is None else x_T
        step_noise, q_noise = [], []
        for _ in range(steps):
            if with_mask:
                q_noise.append(draw())
            step_noise.append(draw())
        return img, torch.stack(step_noise), (torch.stack(q_noise) if with_mask else None)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None,
                      timesteps=None, quantize_denoised=False, mask=None, x0=None, img_callback=None,
                      log_every_t=100, temperature=1.0, noise_dropout=0.0, score_corrector=None,
                      corrector_kwargs=None, unconditional_guidance_scale=1.0,
                      unconditional_conditioning=None, dynamic_threshold=None, ucg_schedule=None):
        """ddim.py:166-262"""
        if ddim_use_original_steps:
            # the reference's own p_sample_ddim reads `self.model.ddim_sigmas_for_original_num_steps` here (ddim.py:324-327), a
            # buffer it registered on the SAMPLER (ddim.py:84-91): with LatentDiffusion as the model this path raises
            # AttributeError in the reference itself, so there is no behaviour to reproduce
            raise NotImplementedError("DDIMSampler(HIP): ddim_use_original_steps=True is not runnable in the reference either "
                                      "(ddim.py:324-327 reads a buffer the model does not have)")
        dev = torch.device("cuda")
        b = shape[0]
        ts = self.ddim_timesteps
        if timesteps is not None:
            # ddim.py:198-206: sample only the first `subset_end` entries of the DDIM sequence (start from a less noisy state)
            subset_end = int(min(timesteps / ts.shape[0], 1) * ts.shape[0]) - 1
            ts = ts[:subset_end]
        total_steps = ts.shape[0]
        time_range = np.flip(ts)
        use_cfg = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.0)
        if total_steps == 0:
            # `timesteps` <= one DDIM interval: the reference's loop runs zero iterations and returns x_T (ddim.py:198-262)
            img = (self._drawer(tuple(shape))() if x_T is None else x_T.detach().float().cpu()).to(dev).contiguous()
            return img, {"x_inter": [img], "pred_x0": [img]}

        # RNG contract R: x_T first, then the per-step draws, all from the host generator in the
        # reference's order; the per-step draws are streamed (see _NoiseFeed)
        draw = self._drawer(tuple(shape))
        img_h = draw() if x_T is None else x_T.detach().float().cpu()
        img = img_h.to(dev).contiguous()
        # device tables in loop order (i = 0 is the noisiest step, index = total_steps - 1)
        order = [total_steps - i - 1 for i in range(total_steps)]
        coef = torch.zeros(total_steps, 8)
        coef[:, :5] = self.ddim_coef[order]
        coef[:, 5] = float(unconditional_guidance_scale)
        coef[:, 6] = 1.0 if use_cfg else 0.0
        coef = coef.to(dev)
        nrep = 2 if use_cfg else 1
        t_tab = torch.from_numpy(np.ascontiguousarray(time_range)).float()[:, None].repeat(1, nrep * b).to(dev).contiguous()

        if mask is not None:
            assert x0 is not None
            mask_d = mask.float().to(dev).expand(shape).contiguous()
            x0_d = x0.float().to(dev).contiguous()
            tr = torch.from_numpy(np.ascontiguousarray(time_range - 0))
            blend_coef = torch.stack([self.sqrt_alphas_cumprod[tr], self.sqrt_one_minus_alphas_cumprod[tr]],
                                     1).contiguous().to(dev)  # [S, 2] = {sqrt(abar_t), sqrt(1 - abar_t)}

        cfg_fused = use_cfg and hasattr(self.model, "apply_model_cfg")
        prepared = self.model.prepare_cfg(cond, unconditional_conditioning) \
            if cfg_fused and hasattr(self.model, "prepare_cfg") else None

        # The captured step graph, its static input buffers and the batched conditioning live on the UNet
        # module and are reused by the next sampling run of the same geometry (text_to_audio in a loop):
        # that run skips the eager first step and the re-capture (~0.1 s per job).  New conditioning is
        # copied INTO the captured tensors and the cross-attention K/V projections are recomputed in
        # place; invalidate_packed() (weights changed) drops the cache.
        # The graph's per-step inputs are selected ON THE DEVICE: a step counter indexes the coefficient table and the noise
        # buffer (aldm_ddim_step_indexed, in place on x) and the graph's last node advances the counter and writes the next
        # step's timestep row (aldm_step_advance) — between two replays the host issues nothing but a stream wait on the
        # noise upload event, once per 8 steps (round 2 issued three copies per step from the host).
        unet = getattr(getattr(self.model, "model", None), "diffusion_model", None)
        can_cache = (prepared is not None and mask is None and self.use_graph and hasattr(unet, "refresh_context_kv")
                     and os.environ.get("ALDM_NO_GRAPH_CACHE", "0") != "1")
        key = None
        if can_cache:
            key = (tuple(shape), total_steps, tuple(tuple(c.shape) for c in prepared["ctxs"]),
                   None if prepared["y"] is None else tuple(prepared["y"].shape))
        ent = unet._graph_cache.get(key) if can_cache else None
        if ent is not None and ent.get("kv") is None:
            # the entry never got as far as recording the K/V buffers its graph reads (e.g. an interrupted first run)
            drop_graph_entries(unet._graph_cache)
            ent = None
        if ent is not None:
            for dst, src in zip(ent["prepared"]["ctxs"] + ent["prepared"]["masks"],
                                prepared["ctxs"] + prepared["masks"]):
                dst.copy_(src)
            if prepared["y"] is not None:
                ent["prepared"]["y"].copy_(prepared["y"])
            unet.refresh_context_kv(ent["kv"])  # in place: the captured graph keeps reading these buffers
            ent["x_cur"].copy_(img)
            ent["coef_all"].copy_(coef)
            ent["t_tab"].copy_(t_tab)
        else:
            # static buffers = the graph's inputs
            ent = {"x_cur": img.clone(), "pred_x0": torch.empty_like(img), "t_cur": t_tab[0].clone(),
                   "step_idx": torch.zeros(1, device=dev, dtype=torch.int32), "coef_all": coef.clone(), "t_tab": t_tab.clone(),
                   "noise_all": torch.empty((total_steps,) + tuple(shape), device=dev, dtype=torch.float32),
                   "prepared": prepared}

            def step(e=ent):
                x_c = e["x_cur"]
                if use_cfg:
                    if cfg_fused:
                        eps = self.model.apply_model_cfg(x_c, e["t_cur"], cond, unconditional_conditioning,
                                                         prepared=e["prepared"])
                    else:
                        tl = e["t_cur"][:b].long()
                        e_u = self.model.apply_model(x_c, tl, unconditional_conditioning)
                        e_c = self.model.apply_model(x_c, tl, cond)
                        eps = torch.stack([e_u, e_c]).contiguous()
                else:
                    eps = self.model.apply_model(x_c, e["t_cur"][:b].long(), cond).contiguous()
                ops.ddim_step_indexed(x_c, eps, e["noise_all"], e["coef_all"], e["step_idx"], e["pred_x0"])
                ops.step_advance(e["step_idx"], e["t_tab"], e["t_cur"])
            ent["run_step"] = GraphStepper(step, self.use_graph)
            if can_cache:
                drop_graph_entries(unet._graph_cache)  # one geometry at a time: the graph pins its activation pool
                unet._graph_cache[key] = ent
        ent["step_idx"].zero_()
        ent["t_cur"].copy_(ent["t_tab"][0])
        x_cur, pred_x0 = ent["x_cur"], ent["pred_x0"]
        # the per-step draws are streamed by a drawer thread straight into the graph's static noise buffer (see _NoiseFeed)
        # While a run is in flight ONLY the drawer thread may consume torch's default CPU generator (it runs up to three 8-step
        # chunks ahead).  In the reference ANY callback may draw from that generator between two steps, so a run WITH a callback
        # keeps the draws on the launching thread, one step at a time, sequenced with the callback exactly as in the reference —
        # unless the callback declares that it never draws (`callback.uses_rng = False`), which restores the threaded feed
        # (ADVICE r3 + r4: opt-OUT, not opt-in — a silent interleaving gives a different noise stream with no diagnostic;
        # INTEGRATION.md §5; ALDM_NOISE_THREAD=0 forces the unthreaded feed for every run).
        cb_rng = any(getattr(f, "uses_rng", True) for f in (callback, img_callback) if f is not None)
        feed = _NoiseFeed(draw, total_steps, tuple(shape), mask is not None, temperature, dev, noise_buf=ent["noise_all"],
                          **({"threaded": False, "chunk": 1} if cb_rng else {}))
        feed.produce_next()
        qn = feed.qnoise
        run_step = ent["run_step"]
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        try:
            for i, step_t in enumerate(time_range):
                index = total_steps - i - 1
                opens_chunk = feed.wait(i)  # current stream waits for the upload of step i's noise chunk
                if mask is not None:
                    # img = q_sample(x0, ts)*mask + (1-mask)*img   (ddim.py:226-231, ddpm.py:430-436)
                    ops.inpaint_blend(x_cur, x0_d, qn[i], mask_d, blend_coef[i])
                run_step()  # eager the first time, then one HIP-graph replay per step
                if can_cache and ent.get("kv") is None:
                    # after the eager first step: the K/V projections of this run's contexts exist; the cache entry owns
                    # them from here on (the graph captured at the next step reads these very buffers)
                    ent["kv"] = unet.collect_context_kv(ent["prepared"]["ctxs"])
                if opens_chunk and not cb_rng:
                    feed.produce_next()  # (unthreaded mode only) draw + upload the NEXT chunk while the GPU works on this one
                if callback:
                    callback(i)
                if img_callback:
                    img_callback(pred_x0, i)
                if index % log_every_t == 0 or index == total_steps - 1:
                    intermediates["x_inter"].append(x_cur.clone())
                    intermediates["pred_x0"].append(pred_x0.clone())
        finally:
            feed.close()   # the generator is past the run's last draw before anyone else draws from it
        return x_cur.clone(), intermediates

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False,
                      quantize_denoised=False, temperature=1.0, noise_dropout=0.0, score_corrector=None,
                      corrector_kwargs=None, unconditional_guidance_scale=1.0,
                      unconditional_conditioning=None, dynamic_threshold=None):
        """ddim.py:265-355 — single step with the reference's signature (noise from the host CPU
        generator, like `noise_like` does on a CPU reference run)."""
        b = x.shape[0]
        use_cfg = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.0)
        if use_cfg:
            if hasattr(self.model, "apply_model_cfg"):
                eps = self.model.apply_model_cfg(x, t.float().repeat(2), c, unconditional_conditioning)
            else:
                eps = torch.stack([self.model.apply_model(x, t, unconditional_conditioning),
                                   self.model.apply_model(x, t, c)]).contiguous()
        else:
            eps = self.model.apply_model(x, t, c).contiguous()
        coef = torch.zeros(8)
        coef[:5] = self.ddim_coef[index]
        coef[5] = float(unconditional_guidance_scale)
        coef[6] = 1.0 if use_cfg else 0.0
        if repeat_noise:
            noise = torch.randn((1, *x.shape[1:])).repeat(b, 1, 1, 1)
        else:
            noise = torch.randn(x.shape)
        noise = (noise * temperature).to(x.device)
        x_prev, pred_x0 = ops.ddim_step(x.float().contiguous(), eps, noise.contiguous(), coef.to(x.device))
        return x_prev, pred_x0

    @torch.no_grad()
    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        """ddim.py:434-449: x0 noised to DDIM index t (fast, no exact reconstruction): sqrt(a_t) x0 + sqrt(1 - a_t) noise,
        the noise from the host generator when not given (a CPU reference run's `randn_like`)."""
        if use_original_steps:
            sa, so = self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod
        else:
            sa, so = torch.sqrt(self.ddim_alphas), torch.from_numpy(np.asarray(self.ddim_sqrt_one_minus_alphas))
        if noise is None:
            noise = torch.randn(x0.shape)
        tt = t.detach().cpu().long()
        shape = (x0.shape[0],) + (1,) * (x0.dim() - 1)
        a = sa.float()[tt].reshape(shape).to(x0.device)
        b = so.float()[tt].reshape(shape).to(x0.device)
        return a * x0 + b * noise.to(x0.device)

    @torch.no_grad()
    def decode(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False, callback=None):
        """ddim.py:452-491: run the last `t_start` DDIM steps from x_latent (one p_sample_ddim per step: the eager path over
        the same kernels as ddim_sampling; the step noise comes from the host generator in the reference's order)."""
        if use_original_steps:
            raise NotImplementedError("DDIMSampler(HIP): use_original_steps=True is not runnable in the reference either "
                                      "(ddim.py:324-327 reads a buffer the model does not have)")
        timesteps = self.ddim_timesteps[:t_start]
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        x_dec = x_latent
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((x_latent.shape[0],), int(step), device=x_latent.device, dtype=torch.long)
            x_dec, _ = self.p_sample_ddim(x_dec, cond, ts, index=index, use_original_steps=use_original_steps,
                                          unconditional_guidance_scale=unconditional_guidance_scale,
                                          unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
        return x_dec
