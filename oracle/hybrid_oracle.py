"""
numpy restatement of the reference's ``HybridPretrainedModel`` network — forward and the gradients
autograd produces for it (``collie/model/hybrid_pretrained_matrix_factorization.py``: metadata
layers ``:327-361``, concatenation order ``:430-459``, combined layers + both bias rows ``:363-399``;
``F.leaky_relu`` default slope 0.01).

TEST INFRASTRUCTURE ONLY (see ``oracle/mf_oracle.py`` header for the rule): imported by ``tests/``
alone.  Pinned against ``oracle/torch_path.py:TorchHybrid`` (the same network written with the
reference's torch calls) in ``tests/test_oracle_golden.py``, which in turn is pinned to vectors made
by the reference's OWN class (``tests/golden/ref_hybrid.npz``, generated in the build container by
``tests/golden/make_golden_ref_hybrid.py`` through the import harness): loss, every gradient and
pair scores of three configurations.  The reference's own tests hold no known-answer vector for
this model.
"""
import numpy as np

F32 = np.float32
SLOPE = F32(0.01)


def leaky(x):
    return np.where(x > 0, x, x * SLOPE).astype(F32)


def forward(params, tables, users, items, item_metadata=None, user_metadata=None):
    """params: {'item_metadata_layers': [(W, b), ...] | None, 'user_metadata_layers': ...,
    'combined_layers': [(W, b), ...]} (W as ``nn.Linear.weight`` [out, in]); tables: the four MF
    tables.  Returns (scores [M], cache for ``backward``)."""
    cache = {'users': users, 'items': items}

    def stack(kind, metadata, ids):
        if metadata is None:
            return None
        out = metadata[ids].astype(F32)
        acts = []
        for W, b in params.get(f'{kind}_metadata_layers') or []:
            pre_in = out
            out = leaky(pre_in @ W.T + b)
            acts.append((pre_in, out))
        cache[kind] = acts
        return out

    parts = []
    um = stack('user', user_metadata, users)
    if um is not None:
        parts.append(um)
    parts += [tables['user_embeddings'][users], tables['item_embeddings'][items]]
    im = stack('item', item_metadata, items)
    if im is not None:
        parts.append(im)
    cache['widths'] = [p.shape[1] for p in parts]
    cache['has_user_meta'] = um is not None
    cache['has_item_meta'] = im is not None
    x = np.concatenate(parts, 1).astype(F32)
    acts = []
    for W, b in params['combined_layers'][:-1]:
        out = leaky(x @ W.T + b)
        acts.append((x, out))
        x = out
    W, b = params['combined_layers'][-1]
    acts.append((x, None))
    cache['combined'] = acts
    s = (x @ W.T + b)[:, 0] + tables['user_biases'][users, 0] + tables['item_biases'][items, 0]
    return s.astype(F32), cache


def backward(params, tables, cache, g):
    """Gradients for d loss / d scores = g [M]: {'combined_layers': [(dW, db), ...],
    'item_metadata_layers': [...], 'user_metadata_layers': [...], and dense table gradients
    'user_biases', 'item_biases', 'user_embeddings', 'item_embeddings'} (float64 accumulation)."""
    g = g.astype(np.float64)
    users, items = cache['users'], cache['items']
    out = {}
    for name, ids in (('user_biases', users), ('item_biases', items)):
        d = np.zeros(tables[name].shape, np.float64)
        np.add.at(d[:, 0], ids, g)
        out[name] = d
    dz = g[:, None]
    grads = []
    layers = params['combined_layers']
    for li in range(len(layers) - 1, -1, -1):
        x, act = cache['combined'][li]
        if act is not None:
            dz = dz * np.where(act > 0, 1.0, np.float64(SLOPE))
        W, _ = layers[li]
        grads.append((dz.T @ x.astype(np.float64), dz.sum(0)))
        dz = dz @ W.astype(np.float64)
    out['combined_layers'] = grads[::-1]
    col = 0
    widths = list(cache['widths'])
    names = (['user_meta'] if cache['has_user_meta'] else []) + ['user_embeddings', 'item_embeddings'] + (
        ['item_meta'] if cache['has_item_meta'] else [])
    for name, w in zip(names, widths):
        d = dz[:, col:col + w]
        col += w
        if name in ('user_embeddings', 'item_embeddings'):
            full = np.zeros(tables[name].shape, np.float64)
            np.add.at(full, users if name == 'user_embeddings' else items, d)
            out[name] = full
            continue
        kind = 'user' if name == 'user_meta' else 'item'
        lgrads = []
        mlayers = params.get(f'{kind}_metadata_layers') or []
        for lj in range(len(mlayers) - 1, -1, -1):
            x, act = cache[kind][lj]
            d = d * np.where(act > 0, 1.0, np.float64(SLOPE))
            lgrads.append((d.T @ x.astype(np.float64), d.sum(0)))
            d = d @ mlayers[lj][0].astype(np.float64)
        out[f'{kind}_metadata_layers'] = lgrads[::-1]
    return out
