"""Measured-and-dropped launcher, kept as the timing instrument that produced profiles/r01_timeline_b1_pipelined.txt
(moved out of the product package in round 2): the sequence forward as one hipGraph per (frame, level) task."""
import torch

from m4depth_amd import network_ops as nops


class TaskGraphSequence:
    """A sequence forward as ONE hipGraph PER TASK -- the batched encoder, every (frame, level) of the decoder, the
    final upsample -- replayed by the host on one real HIP stream per frame, ordered by events.

    Why not one big graph (``GraphedSequence``): captured from several streams it has the right dependencies, but
    ROCm 7.2's graph executor starts the next frame's coarse levels only when the current frame reaches level 1
    (profiles/r01_timeline_b1_pipelined.txt), so most of the possible overlap is lost.  With the tasks as separate
    graphs the (frame, level) wavefront runs on real streams: level l of frame t+1 starts the moment level l of
    frame t has signalled its event, and later frames' streams get a higher priority so that their small kernels
    slip into the workgroup slots a chip-filling level-1 convolution of the frame before frees.  Per step the host
    issues ~30 graph launches and ~50 event operations (< 0.5 ms) instead of ~700 kernel launches.

    Streams share nothing but the level state buffers (event ordered); every frame-stream captures into its own
    memory pool, so tasks that run concurrently never alias scratch memory."""

    def __init__(self, model, example, warmup=2, use_priorities=False):
        # MEASURED (tools/debug_taskgraph.py, batch 1): 6.45 ms/step, the same as the single multi-stream graph
        # (6.25 ms) -- the per-task event timeline shows the wavefront overlapping exactly as designed, the step is
        # simply throughput-bound by then.  With stream priorities it is 10.7 ms/step (high-priority queues starve the
        # chip-filling convolutions), hence use_priorities=False.  Kept as the instrument that produced that timeline.
        self.model = model
        self.new_traj = example["new_traj"].clone() if isinstance(example["new_traj"], torch.Tensor) else example["new_traj"]
        self.static = {k: example[k].clone() for k in ("RGB_im", "rot", "trans")}
        self.camera = {k: v.clone() for k, v in example["camera"].items()}
        T = self.seq_len = self.static["RGB_im"].shape[1]
        pyr = model.d_estimator
        L = self.n_lvls = len(pyr.levels)
        if pyr.is_training:
            raise ValueError("TaskGraphSequence replays the inference path")
        # later frames = higher priority (numerically lower); clamp to what the device offers
        def make_stream(t):
            if use_priorities:
                for prio in (min(0, 1 - t), -1 if t >= 2 else 0, 0):
                    try:
                        return torch.cuda.Stream(priority=prio)
                    except Exception:
                        continue
            return torch.cuda.Stream()
        self.streams = [make_stream(t) for t in range(T)]
        self.main = torch.cuda.Stream()
        pyr._streams = self.streams                      # the eager warm-up allocates per-stream scratch on the same streams
        self.main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.main):
            for _ in range(warmup):
                model([self._samples(), self.camera])
        torch.cuda.synchronize()

        nt = torch.unbind(self.new_traj, dim=1) if isinstance(self.new_traj, torch.Tensor) else list(self.new_traj.T)
        bsz = self.static["RGB_im"].shape[0]
        # -- task 0: encoder, batched over the frames, + the level-local intrinsics
        self.g_enc = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_enc, stream=self.main):
            stacked = model.encoder(torch.cat([self.static["RGB_im"][:, t] for t in range(T)], dim=0))
            self.f_pyrs = [[lvl[t * bsz:(t + 1) * bsz] for lvl in stacked] for t in range(T)]
            self.cams = [{"f": self.camera["f"] / 2. ** (lvl + 1), "c": self.camera["c"] / 2. ** (lvl + 1)} for lvl in range(L)]
        # -- one task per (frame, level), captured in wavefront order on the frame's stream and pool
        pools = [torch.cuda.graph_pool_handle() for _ in range(T)]
        self.tasks = {}
        ests = [None] * T
        for diag in range(T + L - 1):
            for t in range(max(0, diag - L + 1), min(T, diag + 1)):
                l = diag - t
                lvl = L - 1 - l
                g = torch.cuda.CUDAGraph()
                prev = None if ests[t] is None else dict(ests[t][-1])
                with torch.cuda.graph(g, pool=pools[t], stream=self.streams[t]):
                    est = pyr.levels[lvl](self.f_pyrs[t][lvl], prev, self.static["rot"][:, t], self.static["trans"][:, t],
                                          self.cams[lvl], nt[t])
                ests[t] = [est] if ests[t] is None else ests[t] + [est]
                self.tasks[(t, lvl)] = g
        self.estimates = [e[::-1] for e in ests]
        h, w = self.static["RGB_im"].shape[2:4]
        self.g_out = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_out, stream=self.main):
            self.depth = nops.resize_nearest(self.estimates[-1][0]["depth"], h, w)
        self.ev_enc = torch.cuda.Event()
        self.ev = {k: torch.cuda.Event() for k in self.tasks}
        model.last_estimates = self.estimates
        torch.cuda.synchronize()

    def _samples(self):
        nt = torch.unbind(self.new_traj, dim=1) if isinstance(self.new_traj, torch.Tensor) else list(self.new_traj.T)
        return [{"RGB_im": self.static["RGB_im"][:, t], "rot": self.static["rot"][:, t],
                 "trans": self.static["trans"][:, t], "new_traj": nt[t]} for t in range(self.seq_len)]

    def __call__(self, data=None):
        caller = torch.cuda.current_stream()
        main = self.main
        main.wait_stream(caller)
        with torch.cuda.stream(main):
            if data is not None:
                for k in self.static:
                    if data[k].data_ptr() != self.static[k].data_ptr():
                        self.static[k].copy_(data[k], non_blocking=True)
                for k in self.camera:
                    if data["camera"][k].data_ptr() != self.camera[k].data_ptr():
                        self.camera[k].copy_(data["camera"][k], non_blocking=True)
            self.g_enc.replay()
            self.ev_enc.record(main)
        T, L = self.seq_len, self.n_lvls
        for st in self.streams:
            st.wait_event(self.ev_enc)
        for diag in range(T + L - 1):
            for t in range(max(0, diag - L + 1), min(T, diag + 1)):
                lvl = L - 1 - (diag - t)
                st = self.streams[t]
                with torch.cuda.stream(st):
                    if t > 0:
                        st.wait_event(self.ev[(t - 1, lvl)])
                    self.tasks[(t, lvl)].replay()
                    self.ev[(t, lvl)].record(st)
        for t in range(T):
            main.wait_event(self.ev[(t, 0)])              # join: the finest level is every stream's last task
        with torch.cuda.stream(main):
            self.g_out.replay()
        caller.wait_stream(main)
        self.model.step_counter += 1
        return self.depth
