"""CPU stand-in for the HIP backend, built on the oracle.  TEST-ONLY (see oracle/cheb_oracle.py)."""
import numpy as np
import torch

from oracle import cheb_oracle as orc


def _csr_np(op, V=None):
    if op is None:   # K = 1 (dense_mix): the operator is never applied
        return np.zeros(V + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32)
    return op.rowptr.cpu().numpy(), op.colind.cpu().numpy(), op.values.cpu().numpy()


class OracleBackend:
    name = "oracle"

    def spmm(self, op, x, alpha=1.0, z=None, beta=0.0, z2=None, gamma=0.0, out=None):
        rp, ci, va = _csr_np(op)
        y = alpha * orc.remap_f64(rp, ci, va, op.shape, x.detach().float().numpy())
        if z is not None:
            y = y + beta * z.detach().double().numpy()
        if z2 is not None:
            y = y + gamma * z2.detach().double().numpy()
        y = torch.from_numpy(y).to(x.dtype)
        return y if out is None else out.copy_(y)

    def cheb_basis(self, op, x, K):
        rp, ci, va = _csr_np(op)
        L = orc._csr64(rp, ci, va, op.shape)
        T = orc.cheb_basis_f64(L, x.detach().float().numpy(), K)
        if K <= 1:
            return torch.empty((0,) + tuple(x.shape), dtype=x.dtype)
        return torch.from_numpy(np.stack(T[1:])).to(x.dtype)

    def relu_bwd(self, dy, y):
        return torch.where(y > 0, dy, torch.zeros_like(dy))

    def cheb_fwd(self, op, x, w, bias, relu=False):
        rp, ci, va = _csr_np(op, x.shape[1])
        y = orc.cheb_forward_f64(
            rp, ci, va, x.detach().float().numpy(), w.detach().float().numpy(),
            None if bias is None else bias.detach().float().numpy(),
        )
        if relu:
            y = np.maximum(y, 0.0)
        K = w.shape[1]
        T = self.cheb_basis(op, x, K) if K > 1 else None
        return torch.from_numpy(y).to(x.dtype), T

    def cheb_fwd_res(self, op, x, w, bias, scale, res, out=None):
        y, T = self.cheb_fwd(op, x, w, bias)
        y = (y.double() * scale.detach().double() + res.detach().double()).to(x.dtype)
        return (y if out is None else out.copy_(y)), T

    def cheb_bwd_res(self, op, x, T, w, dy, need_dx, need_dw, scale=None, dx_add=None):
        dx, dw, db = self.cheb_bwd(op, x, T, w, dy, need_dx, need_dw, need_dw)
        if dx is not None:
            d = dx.double()
            if scale is not None:
                d = d * scale.detach().double()
            if dx_add is not None:
                d = d + dx_add.detach().double().reshape(d.shape)
            dx = d.to(x.dtype)
        return dx, dw, db

    def rezero_param_grads(self, w, bias, dw_raw, db_raw, scale):
        s = scale.detach().double()
        ds = (w.detach().double() * dw_raw.double()).sum()
        if bias is not None:
            ds = ds + (bias.detach().double() * db_raw.double()).sum()
        return ((s * dw_raw.double()).to(w.dtype), None if bias is None else (s * db_raw.double()).to(w.dtype),
                ds.reshape(scale.shape).to(scale.dtype))

    def cheb_bwd(self, op, x, T, w, dy, need_dx, need_dw, need_db):
        rp, ci, va = _csr_np(op, x.shape[1])
        dx, dw, db = orc.cheb_backward_f64(
            rp, ci, va, x.detach().float().numpy(), w.detach().float().numpy(),
            dy.detach().float().numpy(), has_bias=True,
        )
        return (
            torch.from_numpy(dx).to(x.dtype) if need_dx else None,
            torch.from_numpy(dw).to(w.dtype) if need_dw else None,
            torch.from_numpy(db).to(w.dtype) if need_db else None,
        )

    def rezero_fwd(self, c, r, w, out=None):
        y = (w.double() * c.double() + r.double()).to(c.dtype)
        return y if out is None else out.copy_(y)

    def rezero_bwd(self, g, c, w, need_c):
        gc = (w.double() * g.double()).to(g.dtype) if need_c else None
        gw = (g.double() * c.double()).sum().reshape(w.shape).to(w.dtype)
        return gc, gw

    def maxval_pool_fwd(self, op, x):
        rp, ci, va = _csr_np(op)
        y, sel = orc.maxval_pool_np(rp, ci, va, x.detach().float().numpy())
        return torch.from_numpy(y).to(x.dtype), torch.from_numpy(sel)

    def maxval_pool_bwd(self, op, dy, sel):
        return torch.from_numpy(orc.maxval_pool_backward_np(sel.numpy(), op.shape[1], dy.detach().float().numpy())).to(dy.dtype)

    def maxval_unpool_fwd(self, x, sel, v_fine):
        return torch.from_numpy(orc.maxval_unpool_np(sel.numpy(), v_fine, x.detach().float().numpy())).to(x.dtype)

    def maxval_unpool_bwd(self, dy, sel):
        return torch.from_numpy(orc.maxval_unpool_backward_np(sel.numpy(), dy.detach().float().numpy())).to(dy.dtype)
