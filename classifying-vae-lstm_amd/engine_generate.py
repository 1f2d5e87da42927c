"""Sampling on the device: the frame loops of generate_sample (cl_vae/model.py:9-42, cl_vrnn/model.py:9-60) for N sequences
at once, and the stateful single-step sub-models of cl_vrnn (cl_vrnn/model.py:116-162).  Mixins of engine.VaeEngine /
engine.VrnnEngine: they use the engines' buffers, parameters and forward pieces."""
import torch

from . import _lib, ops
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID


def _f(device, *shape):
    return torch.zeros(*shape, dtype=torch.float32, device=device)


class VaeGenerate:
    def generate(self, x_seed, w, nsteps, seed=0, use_graph=True, z_prior=False, persistent=True, xhat_out=None):
        """N independent sequences of `nsteps` frames on the device: the frame loop of cl_vae/model.py:28-41
        (z-encoder on the last frame, z ~ N(mean, exp(lv)) or N(0, 1), decoder on (w, z, frame before last),
        x ~ Bernoulli); eps and u come from the Philox streams 0 / 1 at step = frame index.  x_seed [N,D], w [N,C] device
        tensors.  persistent=True (default where the shapes allow): the whole loop is ONE kernel, a workgroup per
        sequence (csrc/vae_generate.hip; any N); otherwise the layer chain captured once as a hipGraph and replayed per
        frame (N <= batch size).  Same noise, same samples either way."""
        cfg, d = self.cfg, self.device
        N, D, L = int(x_seed.shape[0]), cfg['D'], cfg['L']
        if persistent and cfg['H'] > 0 and ops.vae_generate_supported(D, cfg['H'], L, cfg['C']):
            P = self.P
            f = dict(dtype=torch.float32, device=d)
            Xs = torch.zeros(N, nsteps, D, **f)
            ops.vae_generate(N, nsteps, D, cfg['H'], L, cfg['C'], cfg['use_x_prev'], z_prior, seed,
                             x_seed.to(**f).contiguous(), w.to(**f).contiguous(), P.p('h/kernel'), P.p('h/bias'),
                             P.p('zargs/kernel'), P.p('zargs/bias'), P.p('decoder_h/kernel'), P.p('decoder_h/bias'),
                             P.p('x_decoded_mean/kernel'), P.p('x_decoded_mean/bias'), Xs, xhat_out)
            return Xs
        if N > self.B:
            raise ValueError("%d sequences exceed the engine's batch size %d" % (N, self.B))
        f = dict(dtype=torch.float32, device=d)
        x_in, hist, x_next = x_seed.to(**f).clone(), x_seed.to(**f).clone(), torch.zeros(N, D, **f)
        eps, u = torch.zeros(N, L, **f), torch.zeros(N, D, **f)
        counter = torch.zeros(1, dtype=torch.int32, device=d)
        Xs = torch.zeros(N, nsteps, D, **f)
        w = w.to(**f).contiguous()

        def frame():
            self.encode_z(x_in, w, N)
            ops.philox_normal(eps, N * L, seed, 0, 0, 0, step_dev=counter)
            if z_prior:
                self.zargs[:N].zero_()
            ops.gauss_fwd(N, L, self.zargs, eps, self.z, L, None)
            self.decode(w, self.z, hist if cfg['use_x_prev'] else None, N, act=ACT_SIGMOID)
            ops.philox_uniform(u, N * D, seed, 0, 1, 0, step_dev=counter)
            ops.bernoulli_sample(N * D, self.logits, u, x_next)
            ops.i32_add(counter, 1)
            hist.copy_(x_in)            # the decoder's history lags the encoder input by one frame
            x_in.copy_(x_next)

        graph = None
        for t in range(nsteps):
            if use_graph and t == 1:
                with ops.Graph() as graph:       # frame 0 ran eagerly and sized every workspace
                    frame()
            if graph is not None:
                graph.launch()
            else:
                frame()
            Xs[:, t].copy_(x_next)
        return Xs


class VrnnGenerate:
    # -- stateful single-step inference (the reference's stateful batch-1 sub-models,
    #    cl_vrnn/model.py:116-162; here for any batch of independent sequences) -------------
    def new_state(self, B):
        d, H = self.device, self.cfg['H']
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=d)
        return dict(B=B, h_enc=z(B, H), c_enc=z(B, H), h_dec=z(B, H), c_dec=z(B, H), gates=z(B, 4 * H),
                    hs=z(B, H), zargs=z(B, 2 * self.cfg['L']), xhat=z(B, self.cfg['D']))

    def encode_w(self, X, B):
        """hW -> Wargs for B windows [B, T*D] (:174-181) -> self.wargs[:B]"""
        cfg, P = self.cfg, self.P
        D, T, C1 = cfg['D'], cfg['T'], cfg['C'] - 1
        ops.gemm(X, P.p('hW/kernel'), self.hW, B, D, T * D, bias=P.p('hW/bias'), act=ACT_RELU, ws=self.ws)
        ops.gemm(self.hW, P.p('Wargs/kernel'), self.wargs, B, 2 * C1, D, bias=P.p('Wargs/bias'), ws=self.ws)

    def _lstm_step(self, name, st, hkey, ckey):
        ops.lstm_seq_fwd(st['B'], 1, st['gates'], None, self.P.p(name + '/recurrent_kernel'), st['hs'], None, None,
                         h0=st[hkey], c0=st[ckey], hT=st[hkey], cT=st[ckey], gate_act=self.gate_act, H=self.cfg['H'])

    def enc_step(self, x, w, st, rec_name='encoder_h'):
        """one encoder-LSTM step on [x_t, w] + the Z heads -> st['zargs'] = [z_mean | z_log_var]"""
        cfg, P, B = self.cfg, self.P, st['B']
        D, H, L, Cn = cfg['D'], cfg['H'], cfg['L'], cfg['C']
        g, ws = ops.gemm, self.ws
        g(x, P.p(rec_name + '/kernel'), st['gates'], B, 4 * H, D, ws=ws)
        g(w, P.rows(P.params, rec_name + '/kernel', D), st['gates'], B, 4 * H, Cn, beta=1.0, bias=P.p(rec_name + '/bias'),
          ws=ws)
        self._lstm_step(rec_name, st, 'h_enc', 'c_enc')
        g(st['hs'], P.p('Zargs/kernel'), st['zargs'], B, 2 * L, H, bias=P.p('Zargs/bias'), ws=ws)

    def dec_step(self, z, xp, w, st):
        """one decoder-LSTM step on [x_{t-1}, z_t, w] + sigmoid head -> st['xhat']"""
        cfg, P, B = self.cfg, self.P, st['B']
        D, H, L, Cn = cfg['D'], cfg['H'], cfg['L'], cfg['C']
        g, ws, off = ops.gemm, self.ws, self.off
        if cfg['use_x_prev']:
            g(xp, P.p('decoder_h/kernel'), st['gates'], B, 4 * H, D, ws=ws)
        g(z, P.rows(P.params, 'decoder_h/kernel', off), st['gates'], B, 4 * H, L,
          beta=1.0 if cfg['use_x_prev'] else 0.0, ws=ws)
        g(w, P.rows(P.params, 'decoder_h/kernel', off + L), st['gates'], B, 4 * H, Cn, beta=1.0,
          bias=P.p('decoder_h/bias'), ws=ws)     # three tiny GEMMs: batch-1 sampling is launch-bound, not flop-bound
        self._lstm_step('decoder_h', st, 'h_dec', 'c_dec')
        g(st['hs'], P.p('X_decoded_mean/kernel'), st['xhat'], B, D, H, bias=P.p('X_decoded_mean/bias'),
          act=ACT_SIGMOID, ws=ws)

    def generate(self, x_seed, w, nsteps, seed=0, use_graph=True, z_prior=False, persistent=True, xhat_out=None):
        """Autoregressive generation of N independent sequences on the device.  persistent=True (default where the
        shapes allow): the whole frame loop is ONE kernel, a workgroup per sequence (csrc/generate.hip); otherwise the
        per-frame chain below, captured once and replayed per frame.  Same Philox noise either way.
        xhat_out [N,S+nsteps,D] (persistent path only) receives every frame's note probabilities."""
        cfg = self.cfg
        if persistent and ops.vrnn_generate_supported(cfg['D'], cfg['H'], cfg['L'], cfg['C']):
            return self._generate_persistent(x_seed, w, nsteps, seed, z_prior, xhat_out)
        return self._generate_frames(x_seed, w, nsteps, seed, use_graph, z_prior)

    def _generate_persistent(self, x_seed, w, nsteps, seed, z_prior, xhat_out):
        cfg, P, d = self.cfg, self.P, self.device
        D, H, L, Cn, off = cfg['D'], cfg['H'], cfg['L'], cfg['C'], self.off
        N, S = int(x_seed.shape[0]), int(x_seed.shape[1])
        Xs = torch.zeros(N, nsteps, D, dtype=torch.float32, device=d)
        rows = lambda name, r: P.rows(P.params, name, r)
        ops.vrnn_generate(N, S, nsteps, D, H, L, Cn, self.gate_act, z_prior, seed, x_seed.contiguous() if S else None,
                          w.contiguous(), P.p('encoder_h/kernel'), rows('encoder_h/kernel', D), P.p('encoder_h/bias'),
                          P.p('encoder_h/recurrent_kernel'), P.p('Zargs/kernel'), P.p('Zargs/bias'),
                          P.p('decoder_h/kernel') if cfg['use_x_prev'] else None, rows('decoder_h/kernel', off),
                          rows('decoder_h/kernel', off + L), P.p('decoder_h/bias'), P.p('decoder_h/recurrent_kernel'),
                          P.p('X_decoded_mean/kernel'), P.p('X_decoded_mean/bias'), Xs, xhat_out)
        return Xs

    def _generate_frames(self, x_seed, w, nsteps, seed=0, use_graph=True, z_prior=False):
        """Batched autoregressive generation on the device (the hot loop of cl_vrnn/model.py:47-59 for N
        independent sequences at once, noise from Philox instead of np.random).
        x_seed [N,S,D] device tensor (teacher-forced frames, S may be 0), w [N,C]; returns Xs [N,nsteps,D].
        One frame = encoder step -> z ~ N(mean, exp(lv)) -> decoder step -> x ~ Bernoulli(x_hat); the chain
        is captured once and replayed per frame with no host synchronisation."""
        cfg, d = self.cfg, self.device
        N, S = int(x_seed.shape[0]), int(x_seed.shape[1])
        D, L = cfg['D'], cfg['L']
        f = dict(dtype=torch.float32, device=d)
        st = self.new_state(N)
        x_prev, x_next = torch.zeros(N, D, **f), torch.zeros(N, D, **f)
        eps, u, z = torch.zeros(N, L, **f), torch.zeros(N, D, **f), torch.zeros(N, L, **f)
        counter = torch.zeros(1, dtype=torch.int32, device=d)
        Xs = torch.zeros(N, nsteps, D, **f)
        w = w.contiguous()

        def frame():
            self.enc_step(x_prev, w, st)
            ops.philox_normal(eps, N * L, seed, 0, 0, 0, step_dev=counter)
            if z_prior:
                st['zargs'].zero_()
            ops.gauss_fwd(N, L, st['zargs'], eps, z, L, None)
            self.dec_step(z, x_prev if cfg['use_x_prev'] else None, w, st)
            ops.philox_uniform(u, N * D, seed, 0, 1, 0, step_dev=counter)
            ops.bernoulli_sample(N * D, st['xhat'], u, x_next)
            ops.i32_add(counter, 1)
            x_prev.copy_(x_next)

        if S == 0:
            x_prev.zero_()
        graph = None
        for t in range(S + nsteps):
            if t < S:
                x_prev.copy_(x_seed[:, t])
            if use_graph and t == 1:
                with ops.Graph() as graph:       # step 0 ran eagerly and sized every workspace
                    frame()
            if graph is not None:
                graph.launch()
            else:
                frame()
            if t >= S:
                Xs[:, t - S].copy_(x_next)
        return Xs
