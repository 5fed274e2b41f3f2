"""
Host-side data objects that produce the batch format the hot path consumes.

The reference's data layer (``collie/interactions/{datasets,samplers,dataloaders}.py``) is out of
scope as a kernel; what *is* part of the drop-in boundary is the batch
format, reproduced here exactly:

    ((users int32 [B], pos_items int32 [B]), neg_items int64 [B, K])

(``datasets.py:382-388,439-443``: scipy COO ``row``/``col`` are int32, ``np.random.randint`` gives
int64).  Only what the MF path needs is provided: an ``Interactions`` wrapper over a COO matrix, a
one-``__getitem__``-per-batch loader with approximate negative sampling
(``ApproximateNegativeSamplingInteractionsDataLoader``, ``samplers.py:11-64``) and the default
``InteractionsDataLoader`` whose exact negative sampling is vectorised here (rejection against a
sorted (user, item) key array instead of the reference's per-sample Python loop,
``datasets.py:399-426``).
"""
import math
from typing import Iterable, Optional

import numpy as np
import torch
from scipy.sparse import coo_matrix


class Interactions:
    """Implicit user-item interactions (``datasets.py:196-445``)."""

    def __init__(self,
                 mat=None,
                 users: Optional[Iterable[int]] = None,
                 items: Optional[Iterable[int]] = None,
                 ratings: Optional[Iterable[int]] = None,
                 num_negative_samples: int = 10,
                 allow_missing_ids: bool = False,
                 remove_duplicate_user_item_pairs: bool = True,
                 num_users='infer',
                 num_items='infer',
                 check_num_negative_samples_is_valid: bool = True,
                 max_number_of_samples_to_consider: int = 200,
                 seed: Optional[int] = None):
        if mat is None:
            if users is None or items is None:
                raise ValueError('Either ``mat`` or both ``users`` and ``items`` must be given')
            users = np.asarray(users)
            items = np.asarray(items)
            if len(users) != len(items):
                raise ValueError('Lengths of ``users`` and ``items`` must be equal')
            if ratings is None:
                ratings = np.ones(len(users))
            ratings = np.asarray(ratings)
            keep = ratings != 0      # 0 ratings are ignored for implicit data (datasets.py:293-302)
            users, items, ratings = users[keep], items[keep], ratings[keep]
            if num_users == 'infer':
                num_users = int(users.max()) + 1
            if num_items == 'infer':
                num_items = int(items.max()) + 1
            if not allow_missing_ids:
                if len(np.unique(users)) != num_users or len(np.unique(items)) != num_items:
                    raise ValueError('``users`` / ``items`` are missing ids; pass '
                                     '``allow_missing_ids=True`` to allow this')
            mat = coo_matrix((ratings, (users, items)), shape=(num_users, num_items))
        else:
            mat = coo_matrix(mat)
        if remove_duplicate_user_item_pairs:
            key = mat.row.astype(np.int64) * mat.shape[1] + mat.col
            _, first = np.unique(key, return_index=True)
            if len(first) != len(key):
                first.sort()
                mat = coo_matrix((mat.data[first], (mat.row[first], mat.col[first])), shape=mat.shape)
        self.mat = coo_matrix((mat.data, (mat.row.astype(np.int32), mat.col.astype(np.int32))),
                              shape=mat.shape)
        self.num_users, self.num_items = self.mat.shape
        self.num_interactions = self.mat.nnz
        assert num_negative_samples >= 1
        self.num_negative_samples = num_negative_samples
        self.max_number_of_samples_to_consider = max_number_of_samples_to_consider
        self.seed = seed if seed is not None else int(np.random.randint(0, 2 ** 31 - 1))
        self._rng = np.random.default_rng(self.seed)
        if check_num_negative_samples_is_valid:
            most = int(np.bincount(self.mat.row).max()) if self.mat.nnz else 0
            assert num_negative_samples < self.num_items - most, (
                '``num_negative_samples`` must be less than {}!'.format(self.num_items - most))
        self._pos_keys = None

    def __len__(self):
        return self.num_interactions

    def _positive_keys(self):
        if self._pos_keys is None:
            self._pos_keys = np.sort(self.mat.row.astype(np.int64) * self.num_items + self.mat.col)
        return self._pos_keys

    def _negative_sample(self, user_id):
        """[len(user_id), K] int64 negatives (``datasets.py:390-445``)."""
        user_id = np.atleast_1d(user_id)
        size = (len(user_id), self.num_negative_samples)
        neg = self._rng.integers(0, self.num_items, size=size, dtype=np.int64)
        if self.max_number_of_samples_to_consider > 0:
            keys = self._positive_keys()
            base = user_id.astype(np.int64)[:, None] * self.num_items
            tries = max(1, self.max_number_of_samples_to_consider // self.num_negative_samples)
            for _ in range(tries):
                k = (base + neg).ravel()
                pos = np.searchsorted(keys, k)
                pos[pos >= len(keys)] = len(keys) - 1
                clash = (keys[pos] == k).reshape(size)
                if size[1] > 1:
                    # a candidate already in the sample's negative list is rejected too
                    # (``datasets.py:408-411``): mark every repeat after its first occurrence
                    order = np.argsort(neg, axis=1, kind='stable')
                    srt = np.take_along_axis(neg, order, axis=1)
                    rep = np.zeros(size, dtype=bool)
                    rep[:, 1:] = srt[:, 1:] == srt[:, :-1]
                    dup = np.zeros(size, dtype=bool)
                    np.put_along_axis(dup, order, rep, axis=1)
                    clash |= dup
                if not clash.any():
                    break
                neg[clash] = self._rng.integers(0, self.num_items, size=int(clash.sum()))
        return neg

    def __getitem__(self, index):
        user_id = self.mat.row[index]
        item_id = self.mat.col[index]
        neg = self._negative_sample(user_id)
        if np.ndim(index) == 0:
            neg = neg[0]
        return (user_id, item_id), neg

    def __repr__(self):
        return (f'Interactions object with {self.num_interactions} interactions between '
                f'{self.num_users} users and {self.num_items} items, returning '
                f'{self.num_negative_samples} negative samples per interaction.')


class ExplicitInteractions:
    """Explicit user-item ratings (``datasets.py:448-562``): ``__getitem__`` returns the non-nested
    ``(user ids, item ids, ratings)`` tuple that marks a batch as explicit."""

    def __init__(self, mat=None, users=None, items=None, ratings=None,
                 allow_missing_ids: bool = False, remove_duplicate_user_item_pairs: bool = True,
                 num_users='infer', num_items='infer'):
        if mat is None and ratings is None:
            raise ValueError('Ratings must be provided to ``ExplicitInteractions`` with ``mat`` or '
                             '``ratings`` - both cannot be ``None``!')
        if mat is None:
            users, items, ratings = np.asarray(users), np.asarray(items), np.asarray(ratings)
            if len(users) != len(items) or len(users) != len(ratings):
                raise ValueError('Lengths of ``users``, ``items`` and ``ratings`` must be equal')
            if num_users == 'infer':
                num_users = int(users.max()) + 1
            if num_items == 'infer':
                num_items = int(items.max()) + 1
            if not allow_missing_ids:
                if len(np.unique(users)) != num_users or len(np.unique(items)) != num_items:
                    raise ValueError('``users`` / ``items`` are missing ids; pass '
                                     '``allow_missing_ids=True`` to allow this')
            mat = coo_matrix((ratings, (users, items)), shape=(num_users, num_items))
        else:
            mat = coo_matrix(mat)
        if remove_duplicate_user_item_pairs:
            key = mat.row.astype(np.int64) * mat.shape[1] + mat.col
            _, first = np.unique(key, return_index=True)
            if len(first) != len(key):
                first.sort()
                mat = coo_matrix((mat.data[first], (mat.row[first], mat.col[first])), shape=mat.shape)
        self.mat = coo_matrix((mat.data, (mat.row.astype(np.int32), mat.col.astype(np.int32))),
                              shape=mat.shape)
        self.num_users, self.num_items = self.mat.shape
        self.num_interactions = self.mat.nnz
        self.min_rating = self.mat.data.min() if self.mat.nnz else None
        self.max_rating = self.mat.data.max() if self.mat.nnz else None
        self.seed = 0

    @property
    def num_negative_samples(self):
        raise AttributeError('``num_negative_samples`` does not exist for explicit datasets.')

    def __len__(self):
        return self.num_interactions

    def __getitem__(self, index):
        return self.mat.row[index], self.mat.col[index], self.mat.data[index]

    def __repr__(self):
        return (f'ExplicitInteractions object with {self.num_interactions} interactions between '
                f'{self.num_users} users and {self.num_items} items, with minimum rating of '
                f'{self.min_rating} and maximum rating of {self.max_rating}.')


class _BaseLoader:
    def __init__(self, interactions, batch_size, shuffle):
        if not isinstance(interactions, (Interactions, ExplicitInteractions)):
            raise ValueError('``interactions`` must be an ``Interactions`` or '
                             '``ExplicitInteractions`` object')
        self.interactions = interactions
        self.batch_size = batch_size
        self.shuffle = shuffle
        self._order = np.arange(len(interactions))
        self._rng = np.random.default_rng(interactions.seed)

    num_users = property(lambda self: self.interactions.num_users)
    num_items = property(lambda self: self.interactions.num_items)
    # AttributeError for explicit data, so that ``hasattr(loader, 'num_negative_samples')`` is False
    num_negative_samples = property(lambda self: self.interactions.num_negative_samples)
    num_interactions = property(lambda self: self.interactions.num_interactions)
    mat = property(lambda self: self.interactions.mat)

    def __len__(self):
        return math.ceil(len(self.interactions) / self.batch_size)

    def __iter__(self):
        if self.shuffle:
            self._rng.shuffle(self._order)
        for start in range(0, len(self.interactions), self.batch_size):
            idxs = self._order[start:start + self.batch_size]
            if isinstance(self.interactions, ExplicitInteractions):
                users, items, ratings = self.interactions[idxs]
                yield (torch.from_numpy(np.ascontiguousarray(users)),
                       torch.from_numpy(np.ascontiguousarray(items)),
                       torch.from_numpy(np.ascontiguousarray(ratings)))
                continue
            (users, items), neg = self.interactions[idxs]
            yield ((torch.from_numpy(np.ascontiguousarray(users)),
                    torch.from_numpy(np.ascontiguousarray(items))),
                   torch.from_numpy(neg))


class InteractionsDataLoader(_BaseLoader):
    """Default loader (``dataloaders.py:70-173``): exact negative sampling unless disabled."""

    def __init__(self, interactions=None, mat=None, users=None, items=None, ratings=None,
                 batch_size: int = 1024, shuffle: bool = False, **kwargs):
        if interactions is None:
            interactions = Interactions(mat=mat, users=users, items=items, ratings=ratings, **kwargs)
        super().__init__(interactions, batch_size, shuffle)


class ApproximateNegativeSamplingInteractionsDataLoader(_BaseLoader):
    """Batch-at-once approximate negative sampling (``dataloaders.py:176-294``)."""

    def __init__(self, interactions=None, mat=None, users=None, items=None, ratings=None,
                 batch_size: int = 1024, shuffle: bool = False, **kwargs):
        if interactions is None:
            interactions = Interactions(mat=mat, users=users, items=items, ratings=ratings,
                                        max_number_of_samples_to_consider=0, **kwargs)
        else:
            interactions.max_number_of_samples_to_consider = 0
        super().__init__(interactions, batch_size, shuffle)


class DeviceInteractionsDataLoader(_BaseLoader):
    """
    The loaders above with the data resident in HBM and every batch assembled on the device
    (``collie_sample_batch``): same constructor arguments, same batch format
    ``((users int32 [B], pos int32 [B]), neg int64 [B, K])``, same sampling rules
    (``max_number_of_samples_to_consider == 0``: approximate, ``datasets.py:427-443``; otherwise the
    exact rejection rule of ``datasets.py:399-426`` against the user's positives), but the tensors
    it yields already live on ``device``, so a training epoch never waits for the host: shuffling is
    one device permutation per epoch (``samplers.py:43-44``) and the negatives come from a
    counter-based Philox stream keyed by ``interactions.seed``.
    There is no CPU fallback: ``device`` must be a HIP device.
    """

    def __init__(self, interactions=None, mat=None, users=None, items=None, ratings=None,
                 batch_size: int = 1024, shuffle: bool = False, device='cuda', **kwargs):
        if interactions is None:
            interactions = Interactions(mat=mat, users=users, items=items, ratings=ratings, **kwargs)
        super().__init__(interactions, batch_size, shuffle)
        from collie_amd import _lib
        self.explicit = isinstance(interactions, ExplicitInteractions)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.CollieHipError('DeviceInteractionsDataLoader samples on a HIP device only')
        mat = interactions.mat
        self._rows = torch.from_numpy(np.ascontiguousarray(mat.row, dtype=np.int32)).to(self.device)
        self._cols = torch.from_numpy(np.ascontiguousarray(mat.col, dtype=np.int32)).to(self.device)
        self._indptr = self._indices = None
        if self.explicit:
            self._ratings = torch.from_numpy(np.ascontiguousarray(mat.data)).to(self.device)
        elif interactions.max_number_of_samples_to_consider > 0:
            csr = mat.tocsr()
            csr.sort_indices()
            self._indptr = torch.from_numpy(csr.indptr.astype(np.int64)).to(self.device)
            self._indices = torch.from_numpy(csr.indices.astype(np.int32)).to(self.device)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(int(interactions.seed))
        self.epoch = 0

    def __iter__(self):
        from collie_amd import ops
        it = self.interactions
        n = len(it)
        order = None
        if self.shuffle:
            order = torch.randperm(n, device=self.device, generator=self._gen)
        base = self.epoch * n
        self.epoch += 1
        for start in range(0, n, self.batch_size):
            b = min(self.batch_size, n - start)
            if self.explicit:      # no sampling: index plumbing only
                sl = slice(start, start + b) if order is None else order[start:start + b]
                yield self._rows[sl], self._cols[sl], self._ratings[sl]
                continue
            yield ops.sample_batch(self._rows, self._cols, order, start, b, self._indptr,
                                   self._indices, it.num_items, it.num_negative_samples,
                                   it.max_number_of_samples_to_consider, it.seed, base)
