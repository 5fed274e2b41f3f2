"""
Generates ``tests/golden/ref_hybrid.npz`` from the REFERENCE'S OWN ``HybridPretrainedModel``
(``collie/model/hybrid_pretrained_matrix_factorization.py``, imported verbatim through
``tests/golden/ref_harness.py``) on top of the reference's ``MatrixFactorizationModel``.  Build
container only.  Usage: ``python tests/golden/make_golden_ref_hybrid.py``.

Per case: the state_dict, one implicit batch, ``calculate_loss(batch)`` (base_pipeline.py:582-654
through the model's ``forward`` :401-464), the gradient of every trainable tensor after
``loss.backward()`` and the scores of a list of (user, item) pairs in eval mode.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ref_harness.install()
from collie.model.matrix_factorization import MatrixFactorizationModel                     # noqa: E402
from collie.model.hybrid_pretrained_matrix_factorization import HybridPretrainedModel       # noqa: E402


class Loader(list):
    def __init__(self, num_users, num_items, num_negative_samples):
        super().__init__()
        self.num_users, self.num_items = num_users, num_items
        self.num_negative_samples = num_negative_samples


def case(out, tag, U, N, D, K, B, item_cols, user_cols, seed, **kw):
    rng = np.random.default_rng(seed)
    loader = Loader(U, N, K)
    torch.manual_seed(seed)
    mf = MatrixFactorizationModel(train=loader, embedding_dim=D, lr_scheduler_func=None)
    with torch.no_grad():
        mf.user_embeddings.weight.mul_(8.0)
        mf.item_embeddings.weight.mul_(8.0)
        mf.user_biases.weight.normal_(0, 0.1)
        mf.item_biases.weight.normal_(0, 0.1)
    item_md = torch.tensor(rng.normal(size=(N, item_cols)).astype(np.float32)) if item_cols else None
    user_md = torch.tensor(rng.normal(size=(U, user_cols)).astype(np.float32)) if user_cols else None
    model = HybridPretrainedModel(train=loader, item_metadata=item_md, user_metadata=user_md,
                                  trained_model=mf, lr_scheduler_func=None, **kw)
    model.train()
    for name, v in model.state_dict().items():
        out[f'{tag}|state|{name}'] = v.detach().numpy().copy()
    if item_md is not None:
        out[f'{tag}|item_metadata'] = item_md.numpy()
    if user_md is not None:
        out[f'{tag}|user_metadata'] = user_md.numpy()
    users = torch.tensor(rng.integers(0, U, B).astype(np.int32))
    pos = torch.tensor(rng.integers(0, N, B).astype(np.int32))
    neg = torch.tensor(rng.integers(0, N, (B, K)).astype(np.int64))
    loss = model.calculate_loss(((users, pos), neg))
    loss.backward()
    out[f'{tag}|users'], out[f'{tag}|pos'], out[f'{tag}|neg'] = users.numpy(), pos.numpy(), neg.numpy()
    out[f'{tag}|loss'] = np.float64(loss.detach())
    out[f'{tag}|loss_name'] = np.str_(model.loss_function.__name__)
    for name, p in model.named_parameters():
        if p.requires_grad and p.grad is not None:
            g = p.grad
            out[f'{tag}|grad|{name}'] = (g.to_dense() if g.is_sparse else g).detach().numpy().copy()
        elif p.requires_grad:
            print(tag, 'no gradient reached', name)
    model.eval()
    pu = torch.tensor(rng.integers(0, U, 64))
    pi = torch.tensor(rng.integers(0, N, 64))
    with torch.no_grad():
        out[f'{tag}|pairs_users'], out[f'{tag}|pairs_items'] = pu.numpy(), pi.numpy()
        out[f'{tag}|pairs_scores'] = model(pu, pi).numpy().copy()
    out[f'{tag}|kw'] = np.str_(repr(sorted(kw.items())))
    print(tag, 'loss', float(loss), model.loss_function.__name__,
          'trainable', [n for n, p in model.named_parameters() if p.requires_grad])


def main():
    out = {}
    case(out, 'item_meta_adaptive', U=60, N=50, D=16, K=5, B=96, item_cols=24, user_cols=0, seed=1,
         item_metadata_layers_dims=[32], combined_layers_dims=[32, 16, 8], freeze_embeddings=False,
         loss='adaptive', optimizer='adam', lr=1e-2)
    case(out, 'both_meta_bpr_frozen', U=40, N=70, D=12, K=4, B=64, item_cols=9, user_cols=7, seed=2,
         item_metadata_layers_dims=[10, 6], user_metadata_layers_dims=[5],
         combined_layers_dims=[20], freeze_embeddings=True, loss='bpr', optimizer='sgd', lr=0.1)
    case(out, 'raw_meta_warp', U=50, N=40, D=8, K=6, B=80, item_cols=5, user_cols=0, seed=3,
         combined_layers_dims=[16, 4], freeze_embeddings=False, loss='warp', optimizer='adagrad',
         lr=5e-2)
    np.savez_compressed(os.path.join(HERE, 'ref_hybrid.npz'), **out)
    print('wrote ref_hybrid.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
