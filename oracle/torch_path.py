"""
The reference's CPU op sequence, restated with the same torch calls the reference makes.

TEST INFRASTRUCTURE ONLY (see ``oracle/mf_oracle.py`` header for the rule).  Used by
  * ``tests/golden/make_golden.py`` (build container) to produce the committed golden vectors —
    there the five loss callables are the reference's own ``collie.loss`` functions, imported
    verbatim from ``/root/reference``;
  * ``bench.py``'s ``cpu_baseline`` leg (``kind: "port"``) on the GPU box's host cores, where the
    reference cannot travel and the loss callables below are used instead.

Reference lines followed: ``collie/model/matrix_factorization.py:106-118,143-159`` (tables and
forward), ``collie/model/base/base_pipeline.py:603-634`` (calculate_loss),
``:418-433`` (optimizer construction), ``collie/model/base/trainer.py:458-470`` (step order).
"""
import torch


def _ideal(pos_items, neg_items, metadata, weights):
    # collie/loss/metadata_utils.py:60-71
    total = sum(weights.values())
    if total > 1:
        raise ValueError(f'sum of metadata weights was {total}, must be <=1')
    frac = torch.zeros(pos_items.shape)
    for key, arr in metadata.items():
        arr = arr.squeeze()
        frac = frac + (arr[pos_items.long()] == arr[neg_items.long()]).int() * weights[key]
    return 1.0 - frac


def _reduce(ell, n):
    return (ell.sum() + ell.pow(2).sum()) / n


def t_bpr(pos, neg, num_items=None, positive_items=None, negative_items=None,
          metadata=None, metadata_weights=None):
    c = _ideal(positive_items, negative_items, metadata, metadata_weights) if metadata else 1
    return _reduce(c - torch.sigmoid(pos - neg), len(pos))          # bpr.py:75-89


def t_hinge(pos, neg, num_items=None, positive_items=None, negative_items=None,
            metadata=None, metadata_weights=None):
    c = _ideal(positive_items, negative_items, metadata, metadata_weights) if metadata else 1
    return _reduce(torch.clamp(c - (pos - neg), min=0), len(pos))   # hinge.py:71-85


def _adaptive(base):
    def fn(pos, many_neg, num_items=None, positive_items=None, negative_items=None,
           metadata=None, metadata_weights=None):
        top, idx = torch.max(many_neg, 0)                            # bpr.py:153, hinge.py:157
        if negative_items is not None and positive_items is not None:
            negative_items = negative_items[idx, torch.arange(len(positive_items))]
        return base(pos, top, positive_items=positive_items, negative_items=negative_items,
                    metadata=metadata, metadata_weights=metadata_weights)
    return fn


t_adaptive_bpr = _adaptive(t_bpr)
t_adaptive_hinge = _adaptive(t_hinge)


def t_warp(pos, many_neg, num_items=None, positive_items=None, negative_items=None,
           metadata=None, metadata_weights=None):
    # warp.py:73-154
    K, B = many_neg.shape
    c = 1
    if metadata:
        c = _ideal(positive_items.repeat([K, 1]), negative_items, metadata, metadata_weights).T
    h = c - pos.view(B, 1) + many_neg.T
    viol = (h > 0)
    padded = torch.cat([viol, torch.ones(B, 1, dtype=torch.bool)], dim=1)
    t = torch.argmax(padded.int() * torch.arange(K + 1, 0, -1), 1)
    tries = (t + 1).float()
    w = torch.log(num_items / tries)
    h1 = torch.cat([h, torch.ones(B, 1)], dim=1)
    ell = w * h1[torch.arange(B), t] * (tries <= K).float()
    return _reduce(ell, B)


TORCH_LOSSES = {'bpr': t_bpr, 'hinge': t_hinge, 'adaptive_bpr': t_adaptive_bpr,
                'adaptive_hinge': t_adaptive_hinge, 'warp': t_warp}


class TorchMF(torch.nn.Module):
    """Four ``nn.Embedding`` tables named as the reference names them (matrix_factorization.py:106-117)."""

    def __init__(self, num_users, num_items, dim, sparse, tables=None, y_range=None):
        super().__init__()
        self.user_biases = torch.nn.Embedding(num_users, 1, sparse=sparse)
        self.item_biases = torch.nn.Embedding(num_items, 1, sparse=sparse)
        self.user_embeddings = torch.nn.Embedding(num_users, dim, sparse=sparse)
        self.item_embeddings = torch.nn.Embedding(num_items, dim, sparse=sparse)
        self.y_range = y_range
        with torch.no_grad():
            if tables is None:
                self.user_biases.weight.zero_()
                self.item_biases.weight.zero_()
                self.user_embeddings.weight.normal_(0, 1.0 / (dim * 2.5))
                self.item_embeddings.weight.normal_(0, 1.0 / (dim * 2.5))
            else:
                for name, arr in tables.items():
                    getattr(self, name).weight.copy_(torch.as_tensor(arr))

    def forward(self, users, items):
        preds = (
            torch.mul(self.user_embeddings(users), self.item_embeddings(items)).sum(axis=1)
            + self.user_biases(users).squeeze(1)
            + self.item_biases(items).squeeze(1)
        )
        if self.y_range is not None:
            preds = torch.sigmoid(preds) * (self.y_range[1] - self.y_range[0]) + self.y_range[0]
        return preds

    def calculate_loss(self, batch, loss_fn, metadata=None, metadata_weights=None):
        (users, pos), neg = batch
        users = users.long()
        pos = pos.long()
        neg = torch.transpose(neg, 0, 1).long()
        pos_preds = self(users, pos)
        neg_preds = self(users.repeat(neg.shape[0]), neg.flatten()).view(neg.shape[0], len(users))
        return loss_fn(pos_preds, neg_preds, num_items=self.item_embeddings.num_embeddings,
                       positive_items=pos, negative_items=neg, metadata=metadata,
                       metadata_weights=metadata_weights), pos_preds, neg_preds

    def tables(self):
        return {n: getattr(self, n).weight.detach().numpy().copy()
                for n in ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')}


def make_optimizers(model, optimizer, bias_optimizer, lr, bias_lr, weight_decay=0.0):
    """base_pipeline.py:361-373,418-433,450-477: split on the substring 'bias' in the name."""
    def build(name, params, rate):
        group = [{'params': params, 'lr': rate}]
        if name == 'sparse_adam':
            return torch.optim.SparseAdam(group)
        group[0]['weight_decay'] = weight_decay
        return {'sgd': torch.optim.SGD, 'adagrad': torch.optim.Adagrad,
                'adam': torch.optim.Adam}[name](group)

    named = list(model.named_parameters())
    if bias_optimizer is None:
        return [build(optimizer, [p for _, p in named], lr)]
    return [build(optimizer, [p for n, p in named if 'bias' not in n], lr),
            build(bias_optimizer, [p for n, p in named if 'bias' in n], bias_lr)]


def train_step(model, optimizers, batch, loss_fn, **kw):
    """trainer.py:458-470: zero_grad, loss, backward, step every optimizer."""
    for opt in optimizers:
        opt.zero_grad()
    loss, pos_preds, neg_preds = model.calculate_loss(batch, loss_fn, **kw)
    loss.backward()
    for opt in optimizers:
        opt.step()
    return loss.detach(), pos_preds.detach(), neg_preds.detach()


def explicit_train_step(model, optimizers, batch, loss='mse'):
    """Explicit branch (base_pipeline.py:635-650, :287-292): MSELoss / L1Loss on forward()."""
    users, items, ratings = batch
    loss_fn = torch.nn.MSELoss(reduction='mean') if loss == 'mse' else torch.nn.L1Loss(reduction='mean')
    for opt in optimizers:
        opt.zero_grad()
    preds = model(users.long(), items.long())
    out = loss_fn(preds, ratings.float())
    out.backward()
    for opt in optimizers:
        opt.step()
    return out.detach(), preds.detach()


def eval_users_torch(model, user_ids, n_items, k):
    """metrics.py:60-72,102-107,171: pair expansion, one forward over Bu*N pairs, topk."""
    users = torch.as_tensor(user_ids, dtype=torch.int64).repeat_interleave(n_items)
    items = torch.arange(n_items).repeat(len(user_ids))
    with torch.no_grad():
        preds = model(users, items).view(-1, n_items)
    return preds, preds.topk(k, dim=1).indices


class TorchHybrid(torch.nn.Module):
    """
    The reference's ``HybridPretrainedModel`` network restated with the torch calls it makes
    (``collie/model/hybrid_pretrained_matrix_factorization.py``): table copies ``:243-250``,
    freezing ``:476-484``, metadata layers ``:188-225,327-361``, combined layers ``:297-325``,
    prediction ``:363-399``, forward / concatenation order ``:401-464``.  Module names are the
    reference's (``embeddings.0`` ... ``combined_layer_3``).  Pinned to the reference's own class:
    ``tests/golden/ref_hybrid.npz`` holds loss, gradients and scores computed by
    ``collie.model.hybrid_pretrained_matrix_factorization.HybridPretrainedModel`` itself (imported
    through ``tests/golden/ref_harness.py`` in the build container), and
    ``tests/test_oracle_golden.py`` checks this class against them (the reference's own tests hold
    only statistical thresholds for the model: ``tests/test_model.py``).
    """

    def __init__(self, tables, item_metadata=None, user_metadata=None,
                 item_metadata_layers_dims=None, user_metadata_layers_dims=None,
                 combined_layers_dims=(128, 64, 32), freeze_embeddings=True):
        super().__init__()
        nn = torch.nn
        U, D = tables['user_embeddings'].shape
        N = tables['item_embeddings'].shape[0]
        self.embeddings = nn.Sequential(nn.Embedding(U, D), nn.Embedding(N, D))
        self.biases = nn.Sequential(nn.Embedding(U, 1), nn.Embedding(N, 1))
        with torch.no_grad():
            self.embeddings[0].weight.copy_(torch.as_tensor(tables['user_embeddings']))
            self.embeddings[1].weight.copy_(torch.as_tensor(tables['item_embeddings']))
            self.biases[0].weight.copy_(torch.as_tensor(tables['user_biases']))
            self.biases[1].weight.copy_(torch.as_tensor(tables['item_biases']))
        self.embeddings[0].weight.requires_grad = not freeze_embeddings
        self.embeddings[1].weight.requires_grad = not freeze_embeddings
        self.item_metadata = None if item_metadata is None else torch.as_tensor(item_metadata).float()
        self.user_metadata = None if user_metadata is None else torch.as_tensor(user_metadata).float()
        self.num_items = N

        def stack(kind, metadata, dims):
            if metadata is None:
                return None, None
            if dims is None:
                return None, metadata.shape[1]
            full = [metadata.shape[1]] + list(dims)
            layers = [nn.Linear(full[i - 1], full[i]) for i in range(1, len(full))]
            for i, layer in enumerate(layers):
                nn.init.xavier_normal_(layer.weight)
                self.add_module(f'{kind}_metadata_layer_{i}', layer)
            return layers, dims[-1]

        self.item_metadata_layers, item_out = stack('item', self.item_metadata,
                                                    item_metadata_layers_dims)
        self.user_metadata_layers, user_out = stack('user', self.user_metadata,
                                                    user_metadata_layers_dims)
        dims = [2 * D + (item_out or 0) + (user_out or 0)] + list(combined_layers_dims) + [1]
        self.combined_layers = [nn.Linear(dims[i - 1], dims[i]) for i in range(1, len(dims))]
        for i, layer in enumerate(self.combined_layers):
            nn.init.xavier_normal_(layer.weight)
            self.add_module(f'combined_layer_{i}', layer)

    def _metadata_output(self, metadata, layers, ids):
        out = metadata[ids, :]
        if layers is not None:
            for layer in layers:
                out = torch.nn.functional.leaky_relu(layer(out))
        return out

    def forward(self, users, items):
        parts = []
        if self.user_metadata is not None:
            parts.append(self._metadata_output(self.user_metadata, self.user_metadata_layers, users))
        parts += [self.embeddings[0](users), self.embeddings[1](items)]
        if self.item_metadata is not None:
            parts.append(self._metadata_output(self.item_metadata, self.item_metadata_layers, items))
        out = torch.cat(parts, 1)
        for layer in self.combined_layers[:-1]:
            out = torch.nn.functional.leaky_relu(layer(out))
        out = self.combined_layers[-1](out) + self.biases[0](users) + self.biases[1](items)
        return out.squeeze()

    def calculate_loss(self, batch, loss_fn, metadata=None, metadata_weights=None):
        """base_pipeline.py:603-634 (two forward calls, as the reference makes them)."""
        (users, pos), neg = batch
        users = users.long()
        pos = pos.long()
        neg = torch.transpose(neg, 0, 1).long()
        pos_preds = self(users, pos)
        neg_preds = self(users.repeat(neg.shape[0]), neg.flatten()).view(neg.shape[0], len(users))
        return loss_fn(pos_preds, neg_preds, num_items=self.num_items, positive_items=pos,
                       negative_items=neg, metadata=metadata,
                       metadata_weights=metadata_weights), pos_preds, neg_preds
