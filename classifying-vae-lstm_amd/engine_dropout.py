"""get_model(dropout=p) of cl_vrnn (cl_vrnn/model.py:164,198,227; never set by the reference's scripts): Keras 2.0.0's LSTM input
dropout in training passes, on the generic chain.  A mixin of engine.VrnnEngine."""
from . import ops
from .ops import ACT_MASKPOS


class VrnnDropout:
    def set_dropout_uniforms(self, u_enc, u_dec):
        """The uniforms behind the two LSTMs' input-dropout masks ([B, 4, D + C] and [B, 4, (D) + L + C], m = (u >= p) / (1 - p)) as
        explicit inputs, like eps_W / eps_Z: a training pass without `noise=` uses them."""
        self.u_enc.copy_(u_enc.view_as(self.u_enc)); self.u_dec.copy_(u_dec.view_as(self.u_dec))
        self._masks_given = True

    # -- LSTM(dropout=p), training passes (cl_vrnn/model.py:164,198,227) ---------------------------------------------------
    # Keras 2.0.0 (implementation 0) multiplies the inputs of gate g's projection with a mask m_g [B, input_dim] drawn once per
    # batch -- the same for every time step -- and different for the four gates: z_g = ([x_t, z_t, W] * m_g) . K[:, g] + b_g +
    # h_{t-1} . U[:, g].  u_enc / u_dec [B, 4, input_dim] hold the masks' UNIFORMS (m = (u >= p) / (1 - p), applied by
    # clv_dropout_rows); the per-step rows and the label row of a gate are masked into xm_* / wm_* and multiplied as GEMMs
    # over that gate's column block (ldb = ldc = 4H).  Inference passes (validation, predict, generation) take no dropout.
    def _gate_cols(self, t, gi):
        """1-D / 2-D view of buffer t starting at gate gi's column block (pointer offset; the leading dimension stays 4H)"""
        H = self.cfg['H']
        return t[gi * H:] if t.dim() == 1 else t[:, gi * H:]

    def _forward_dropout(self, X, eps_W, eps_Z, w_true, nll, target):
        cfg, P, B = self.cfg, self.P, self.B
        D, H, L, T, Cn, off = cfg['D'], cfg['H'], cfg['L'], cfg['T'], cfg['C'], self.off
        BT, G4, rate = B * T, 4 * H, self.dropout
        g, ws, gc = ops.gemm, self.ws, self._gate_cols
        self._label_forward(X, eps_W, w_true)              # hW, Wargs, W, the label losses (its unmasked row biases are replaced)
        in_e, in_d = D + Cn, off + L + Cn
        ue, ud = self.u_enc.view(B, 4 * in_e), self.u_dec.view(B, 4 * in_d)
        X2 = X.reshape(BT, D)
        for gi in range(4):
            ops.dropout_rows(BT, T, D, X2, D, ue[:, gi * in_e:], 4 * in_e, rate, self.xm_e[gi], D)
            g(self.xm_e[gi], gc(P.p('encoder_h/kernel'), gi), gc(self.gates_enc, gi), BT, H, D, ldb=G4, ldc=G4, ws=ws)
            ops.dropout_rows(B, 1, Cn, self.W, Cn, ue[:, gi * in_e + D:], 4 * in_e, rate, self.wm_e[gi], Cn)
            g(self.wm_e[gi], gc(P.rows(P.params, 'encoder_h/kernel', D), gi), gc(self.wk_enc, gi), B, H, Cn, ldb=G4, ldc=G4,
              bias=gc(P.p('encoder_h/bias'), gi), ws=ws)
        ops.lstm_seq_fwd(B, T, self.gates_enc, self.wk_enc, P.p('encoder_h/recurrent_kernel'), self.hs_enc, self.cs_enc,
                         self.gates_enc, gate_act=self.gate_act, H=H)
        if self.fuse_latent:
            ops.latent_head_fwd(BT, H, L, self.hs_enc, P.p('Zargs/kernel'), P.p('Zargs/bias'), eps_Z, self.zargs, self.Z,
                                self.xz_ld, self.rowkl)
        else:
            g(self.hs_enc, P.p('Zargs/kernel'), self.zargs, BT, 2 * L, H, bias=P.p('Zargs/bias'), ws=ws)
            ops.gauss_fwd(BT, L, self.zargs, eps_Z, self.Z, self.xz_ld, self.rowkl)
        for gi in range(4):
            ops.dropout_rows(BT, T, off + L, self.XZ, self.xz_ld, ud[:, gi * in_d:], 4 * in_d, rate, self.xm_d[gi], self.xz_ld)
            g(self.xm_d[gi], gc(P.p('decoder_h/kernel'), gi), gc(self.gates_dec, gi), BT, H, off + L, lda=self.xz_ld, ldb=G4,
              ldc=G4, ws=ws)
            ops.dropout_rows(B, 1, Cn, self.W, Cn, ud[:, gi * in_d + off + L:], 4 * in_d, rate, self.wm_d[gi], Cn)
            g(self.wm_d[gi], gc(P.rows(P.params, 'decoder_h/kernel', off + L), gi), gc(self.wk_dec, gi), B, H, Cn, ldb=G4,
              ldc=G4, bias=gc(P.p('decoder_h/bias'), gi), ws=ws)
        ops.lstm_seq_fwd(B, T, self.gates_dec, self.wk_dec, P.p('decoder_h/recurrent_kernel'), self.hs_dec, self.cs_dec,
                         self.gates_dec, gate_act=self.gate_act, H=H)
        self._output_head(target, nll)

    def _backward_dropout(self, X, w_true, eps_W, eps_Z):
        """Everything behind dL/dh_dec (self.dhs) of a training pass with input dropout: both BPTTs, dZ and dW through the
        gates' masks, the latent head, the label path from an explicit dL/dW, every weight gradient.  Plain launches (no
        deferred reductions): this chain is the rarely used one."""
        cfg, P, B = self.cfg, self.P, self.B
        D, H, L, T, Cn, off = cfg['D'], cfg['H'], cfg['L'], cfg['T'], cfg['C'], self.off
        C1, BT, G4, rate = Cn - 1, B * T, 4 * H, self.dropout
        g, ws, gc = ops.gemm, self.ws, self._gate_cols
        in_e, in_d = D + Cn, off + L + Cn
        ue, ud = self.u_enc.view(B, 4 * in_e), self.u_dec.view(B, 4 * in_d)
        Kd, Ke = P.p('decoder_h/kernel'), P.p('encoder_h/kernel')
        # decoder BPTT; dZ = sum_g m_g[z cols] * (dz_g . K_z[:, g]^T); dW (decoder share) likewise from sum_t dz
        ops.lstm_seq_bwd(B, T, P.p('decoder_h/recurrent_kernel'), self.dhs, self.cs_dec, self.gates_dec, self.dzsum_dec,
                         gate_act=self.gate_act, H=H)
        for gi in range(4):
            g(gc(self.gates_dec, gi), gc(P.rows(P.params, 'decoder_h/kernel', off), gi), self.dxg, BT, L, H, tb=True, lda=G4,
              ldb=G4, ldc=self.xz_ld, ws=ws)
            ops.dropout_rows(BT, T, L, self.dxg, self.xz_ld, ud[:, gi * in_d + off:], 4 * in_d, rate, self.dZ, L, beta=float(gi > 0))
            g(gc(self.dzsum_dec, gi), gc(P.rows(P.params, 'decoder_h/kernel', off + L), gi), self.dwg, B, Cn, H, tb=True, lda=G4,
              ldb=G4, ws=ws)
            ops.dropout_rows(B, 1, Cn, self.dwg, Cn, ud[:, gi * in_d + off + L:], 4 * in_d, rate, self.dW, Cn, beta=float(gi > 0))
        # latent head backward -> dL/dh_enc
        if self.fuse_latent:
            ops.latent_head_bwd(BT, H, L, self.hs_enc, P.p('Zargs/kernel'), self.zargs, eps_Z, self.dZ, L,
                                self.kl_weight / BT, self.dhs, P.g('Zargs/kernel'), P.g('Zargs/bias'), ws, defer=None)
        else:
            ops.gauss_bwd(BT, L, self.zargs, eps_Z, self.dZ, L, self.kl_weight / BT, self.dzargs)
            g(self.dzargs, P.p('Zargs/kernel'), self.dhs, BT, H, 2 * L, tb=True, ws=ws)
            self._dense_wgrad('Zargs', self.hs_enc, H, H, 2 * L, BT, self.dzargs, ws, None)
        ops.lstm_seq_bwd(B, T, P.p('encoder_h/recurrent_kernel'), self.dhs, self.cs_enc, self.gates_enc, self.dzsum_enc,
                         gate_act=self.gate_act, H=H)
        for gi in range(4):
            g(gc(self.dzsum_enc, gi), gc(P.rows(P.params, 'encoder_h/kernel', D), gi), self.dwg, B, Cn, H, tb=True, lda=G4,
              ldb=G4, ws=ws)
            ops.dropout_rows(B, 1, Cn, self.dwg, Cn, ue[:, gi * in_e + D:], 4 * in_e, rate, self.dW, Cn, beta=1.0)
        # label path from dL/dW (the LSTMs' share so far; clv_label_bwd adds the label losses' own terms)
        ops.label_bwd(B, Cn, self.wargs, self.wargs[:, C1:], 2 * C1, eps_W, w_true, self.W, self.dW,
                      cfg['w_log_var_prior'], self.class_weight, self.w_kl_weight, 1.0 / B,
                      self.dwargs, self.dwargs[:, C1:], 2 * C1)
        g(self.hW, self.dwargs, P.g('Wargs/kernel'), D, 2 * C1, B, ta=True, ws=ws)
        ops.colsum(self.dwargs, B, 2 * C1, P.g('Wargs/bias'), ws)
        g(self.dwargs, P.p('Wargs/kernel'), self.dhW, B, D, 2 * C1, tb=True, act=ACT_MASKPOS, aux=self.hW, ws=ws)
        g(X.reshape(B, T * D), self.dhW, P.g('hW/kernel'), T * D, D, B, ta=True, ws=ws)
        ops.colsum(self.dhW, B, D, P.g('hW/bias'), ws)
        self.gdot_fresh = False
        # weight gradients of the two LSTMs: per gate the masked inputs' products, the recurrent kernel as usual
        for name, xm, wm, nin, ldx, hs, dz, dzsum in (('encoder_h', self.xm_e, self.wm_e, D, D, self.hs_enc, self.gates_enc, self.dzsum_enc),
                                                      ('decoder_h', self.xm_d, self.wm_d, off + L, self.xz_ld, self.hs_dec, self.gates_dec,
                                                       self.dzsum_dec)):
            for gi in range(4):
                g(xm[gi], gc(dz, gi), gc(P.g(name + '/kernel'), gi), nin, H, BT, ta=True, lda=ldx, ldb=G4, ldc=G4, ws=ws)
                g(wm[gi], gc(dzsum, gi), gc(P.rows(P.grads, name + '/kernel', nin), gi), Cn, H, B, ta=True, ldb=G4, ldc=G4, ws=ws)
            ops.gemm_grouped_tn([dict(A=hs, lda=H, M=H, C=P.g(name + '/recurrent_kernel'), shift=1, zero_period=T)], G4, BT, dz, ws)
            ops.colsum(dzsum, B, G4, P.g(name + '/bias'), ws)
        if not self._head_done:
            self._dense_wgrad('X_decoded_mean', self.hs_dec, H, H, D, BT, self.dlogits, ws, None)
        rq = self._rq()
        if rq is not None:       # (the fused output head may have left its slabs pending; the loss means ride along)
            rq.flush(means=getattr(self, '_loss_terms', None), out=self.scal, skinny=None)
            self._loss_terms = None
