"""
TEST INFRASTRUCTURE: a stand-in for ``collie_amd.sharded.HipEngine`` whose arithmetic is the CPU
oracle (``oracle/mf_oracle.py``), so that the world_size-2 ``gloo`` tests can drive the product's
exchange code (plan / all-to-all of the packed buffers / candidate merge) without a GPU, in the
product's buffer layout.
Never imported by ``collie_amd``.
"""
import numpy as np
import torch

from oracle import mf_oracle as O

F32 = np.float32


def _np(t):
    return t.detach().numpy()


class CpuOracleEngine:
    name = 'oracle'

    # ---- training -------------------------------------------------------------------------------
    # The exchange buffers have the product's layout (csrc/shard_wire.h): per peer, c rows, then
    # ceil(c / D) rows of bias values, then (gradient direction) as many rows of int32 row numbers.
    @staticmethod
    def _blocks(counts, D, extra):
        first, row0 = [0], [0]
        for c in counts:
            first.append(first[-1] + c)
            row0.append(row0[-1] + c + extra * ((c + D - 1) // D))
        return first, row0

    def serve(self, model, rows, peer_counts, step):
        emb, bias = _np(model.emb), _np(model.bias)
        D = emb.shape[1]
        first, row0 = self._blocks(peer_counts, D, 1)
        out = np.zeros((row0[-1], D), F32)
        flat = out.reshape(-1)
        idx = _np(rows)
        for p, c in enumerate(peer_counts):
            r = idx[first[p]:first[p] + c]
            out[row0[p]:row0[p] + c] = emb[r]
            flat[(row0[p] + c) * D:(row0[p] + c) * D + c] = bias[r]
        # looked-up item rows of this step (SparseAdam touches them even without a gradient)
        self._touched = set(int(r) for r in idx if r >= model.U_loc)
        return torch.from_numpy(out)

    def forward_export(self, model, cw, lookup_counts, send_rows, cu, cp, cn, kind, num_items, codes,
                       weights, scale, users_local=False, step=None):
        buf = _np(cw)
        R, D = buf.shape
        first, row0 = self._blocks(lookup_counts, D, 1)
        flat = buf.reshape(-1)
        cbias = np.zeros(R, F32)
        owner_of = np.zeros(R, np.int64)
        dense_of = np.zeros(R, np.int64)
        for p, c in enumerate(lookup_counts):
            cbias[row0[p]:row0[p] + c] = flat[(row0[p] + c) * D:(row0[p] + c) * D + c]
            owner_of[row0[p]:row0[p] + c] = p
            dense_of[row0[p]:row0[p] + c] = np.arange(first[p], first[p] + c)
        tables = {'user_embeddings': buf, 'item_embeddings': buf,
                  'user_biases': cbias.reshape(R, 1), 'item_biases': cbias.reshape(R, 1)}
        if users_local:     # the user rows are this rank's own
            nu = model.U_loc
            tables['user_embeddings'] = _np(model.emb)[:nu]
            tables['user_biases'] = _np(model.bias)[:nu].reshape(nu, 1)
        metadata = {f'm{i}': _np(c) for i, c in enumerate(codes)} or None
        mweights = {f'm{i}': w for i, w in enumerate(weights)} or None
        out = O.calculate_loss(tables, _np(cu), _np(cp), _np(cn), kind, num_items=num_items,
                               metadata=metadata, metadata_weights=mweights)
        s = F32(scale.item())
        gemb = np.zeros((R, D), F32)
        for name in ('item_embeddings',) if users_local else ('user_embeddings', 'item_embeddings'):
            rows, vals = out['grads'][name]
            gemb[rows] += vals * s
        if users_local:     # the user rows never leave: update them here
            rows, vals = out['grads']['user_embeddings']
            hp = model.hparams
            self._rows_update(_np(model.emb), np.asarray(rows, np.int64), vals * s,
                              hp['optimizer'], step, hp['lr'], model.state['emb'])
        gbias = np.zeros(R, F32)
        rows, vals = out['grads']['item_biases']        # d loss / d user_bias is identically 0
        gbias[rows] = vals[:, 0] * s
        # rows that travel: users, positives and every row with a non-zero gradient (the product
        # sends users, positives and the selected negatives); the other looked-up rows reach the
        # owner's update through the stamps of serve()
        send = np.zeros(R, bool)
        if not users_local:
            send[_np(cu)] = True
        send[_np(cp)] = True
        send |= (gemb != 0).any(axis=1) | (gbias != 0)
        srows = _np(send_rows)
        G = len(lookup_counts)
        sel = [np.nonzero(send & (owner_of == p) & (np.arange(R) >= row0[p]) &
                          (np.arange(R) < row0[p] + lookup_counts[p]))[0] for p in range(G)]
        counts = [len(x) for x in sel]
        _, w0 = self._blocks(counts, D, 2)
        wire = np.zeros((w0[-1], D), F32)
        wflat = wire.reshape(-1)
        wint = wflat.view(np.int32)
        for p, rows_p in enumerate(sel):
            c = counts[p]
            extra = (c + D - 1) // D
            wire[w0[p]:w0[p] + c] = gemb[rows_p]
            wflat[(w0[p] + c) * D:(w0[p] + c) * D + c] = gbias[rows_p]
            wint[(w0[p] + c + extra) * D:(w0[p] + c + extra) * D + c] = srows[dense_of[rows_p]]
        return (torch.tensor([out['loss']], dtype=torch.float32), torch.from_numpy(wire),
                torch.tensor(counts + [w0[-1]], dtype=torch.int64))

    def owner_update(self, model, wire, peer_counts, step):
        hp = model.hparams
        emb, bias = _np(model.emb), _np(model.bias)
        D = emb.shape[1]
        buf = _np(wire) if wire is not None and wire.shape[0] else np.zeros((0, D), F32)
        flat = buf.reshape(-1)
        fint = flat.view(np.int32)
        _, w0 = self._blocks(peer_counts, D, 2)
        rows, gemb, gbias = [], [], []
        for p, c in enumerate(peer_counts):              # peer order = requester rank order
            extra = (c + D - 1) // D
            rows.append(fint[(w0[p] + c + extra) * D:(w0[p] + c + extra) * D + c].astype(np.int64))
            gemb.append(buf[w0[p]:w0[p] + c])
            gbias.append(flat[(w0[p] + c) * D:(w0[p] + c) * D + c])
        rows = np.concatenate(rows) if rows else np.zeros(0, np.int64)
        gemb = np.concatenate(gemb) if gemb else np.zeros((0, D), F32)
        gbias = np.concatenate(gbias) if gbias else np.zeros(0, F32)
        have = set(rows.tolist())
        extra_rows = np.array(sorted(r for r in self._touched if r not in have), np.int64)
        rows = np.concatenate([rows, extra_rows])
        gemb = np.concatenate([gemb, np.zeros((len(extra_rows), D), F32)])
        gbias = np.concatenate([gbias, np.zeros(len(extra_rows), F32)])
        if rows.shape[0] == 0:
            return
        for table, vals, algo, lr, state in (
                (emb, gemb, hp['optimizer'], hp['lr'], model.state['emb']),
                (bias, gbias, hp['bias_optimizer'], hp['bias_lr'], model.state['bias'])):
            self._rows_update(table, rows, vals, algo, step, lr, state)

    @staticmethod
    def _rows_update(p, idx, vals, algo, step, lr, state):
        p2 = p.reshape(p.shape[0], -1)
        v = vals.reshape(idx.shape[0], -1).astype(F32)
        urows, inv = np.unique(idx, return_inverse=True)
        g = np.zeros((urows.shape[0], v.shape[1]), F32)
        for e in range(idx.shape[0]):                    # entry order: requester rank order
            g[inv[e]] += v[e]
        st = {k: _np(t).reshape(p2.shape) for k, t in state.items()}
        if algo == 'sparse_adam':
            O.sparse_adam_step(p2, st['exp_avg'], st['exp_avg_sq'], step, urows, g, lr)
        elif algo == 'adagrad':
            O.adagrad_step(p2, st['sum'], step, lr, rows=urows, g=g)
        else:
            O.sgd_step(p2, lr, rows=urows, g=g)

    # ---- evaluation -----------------------------------------------------------------------------
    @staticmethod
    def _scores(utile, ubias, iemb, ibias):
        n, m = utile.shape[0], iemb.shape[0]
        tables = {'user_embeddings': _np(utile), 'item_embeddings': _np(iemb),
                  'user_biases': _np(ubias).reshape(n, 1), 'item_biases': _np(ibias).reshape(m, 1)}
        return O.get_preds(tables, np.arange(n), m)

    def shard_topk(self, utile, ubias, iemb, ibias, stride, offset, k):
        s = self._scores(utile, ubias, iemb, ibias)
        loc = O.topk_ids(s, k)
        return (torch.from_numpy(np.take_along_axis(s, loc, axis=1).astype(F32)),
                torch.from_numpy(loc * stride + offset))

    def merge_topk(self, cand_scores, cand_ids, k):
        n = cand_scores.shape[0]
        cs = _np(cand_scores).reshape(n, -1)
        ci = _np(cand_ids).reshape(n, -1).astype(np.int64)
        out = np.empty((n, k), np.int64)
        for u in range(n):
            order = np.lexsort((ci[u], -cs[u].astype(np.float64)))
            out[u] = ci[u][order[:k]]
        return torch.from_numpy(out)

    def shard_pos_scores(self, utile, ubias, iemb, ibias, stride, offset, indptr, indices):
        s = self._scores(utile, ubias, iemb, ibias)
        ip, ix = _np(indptr), _np(indices)
        pos = np.zeros(max(ix.shape[0], 1), F32)
        for u in range(utile.shape[0]):
            for e in range(ip[u], ip[u + 1]):
                rel = ix[e] - offset
                if rel >= 0 and rel % stride == 0:
                    pos[e] = s[u, rel // stride]
        return torch.from_numpy(pos)

    def thresholds(self, utile, ubias, iemb, ibias, indptr, indices, pos, y_range):
        assert y_range is None
        return (pos, indptr, indices)

    def shard_rank_counts(self, utile, ubias, iemb, ibias, stride, offset, indptr, thr):
        pos, ip, ix = (_np(t) for t in thr)
        s = self._scores(utile, ubias, iemb, ibias)
        gid = np.arange(iemb.shape[0]) * stride + offset
        sig = O._sigmoid32(s)
        counts = np.zeros((utile.shape[0], 3), np.int64)
        for u in range(utile.shape[0]):
            a, b = ip[u], ip[u + 1]
            if a == b:
                continue
            ps, pi = pos[a:b], ix[a:b]
            best = np.lexsort((pi, -ps.astype(np.float64)))[0]
            bs, bi = ps[best], pi[best]
            counts[u, 0] = np.sum((s[u] > bs) | ((s[u] == bs) & (gid < bi)))
            psig = O._sigmoid32(ps)
            counts[u, 1] = sum(int(np.sum(sig[u] < x)) for x in psig)
            counts[u, 2] = sum(int(np.sum(sig[u] <= x)) for x in psig)
        return torch.from_numpy(counts)

    def metrics(self, topk_ids, k, n_users, n_items, indptr, indices, pos, counts, y_range, want):
        want_ap, want_rr, want_auc = want
        ip, ix = _np(indptr), _np(indices)
        ap = rr = auc = None
        if want_ap:
            ids = _np(topk_ids)
            ap = np.zeros(n_users, np.float64)
            for u in range(n_users):
                truth = set(ix[ip[u]:ip[u + 1]].tolist())
                hits, acc = 0, 0.0
                for j in range(k):
                    if int(ids[u, j]) in truth:
                        hits += 1
                        acc += hits / (j + 1.0)
                denom = min(k, len(truth))
                ap[u] = acc / denom if denom else 0.0
            ap = torch.from_numpy(ap)
        if want_rr or want_auc:
            c = _np(counts)
            ps = _np(pos)
        if want_rr:
            rr = np.zeros(n_users, F32)
            for u in range(n_users):
                if ip[u + 1] > ip[u]:
                    rr[u] = F32(1.0) / F32(c[u, 0] + 1)
            rr = torch.from_numpy(rr)
        if want_auc:
            auc = np.zeros(n_users, np.float64)
            for u in range(n_users):
                P = ip[u + 1] - ip[u]
                Nn = n_items - P
                if P == 0 or Nn == 0:
                    continue
                psig = O._sigmoid32(ps[ip[u]:ip[u + 1]])
                lt_pp = sum(int(np.sum(psig < x)) for x in psig)
                le_pp = sum(int(np.sum(psig <= x)) for x in psig)
                lt = c[u, 1] - lt_pp
                le = c[u, 2] - le_pp
                auc[u] = (lt + 0.5 * (le - lt)) / (P * Nn)
            auc = torch.from_numpy(auc)
        return ap, rr, auc
