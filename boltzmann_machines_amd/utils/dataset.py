"""Readers for the two datasets the reference's examples train on (boltzmann_machines/utils/dataset.py
:10-72 `load_mnist`, `load_cifar10`; :74-130 `im_flatten` / `im_unflatten`).  Same file layouts under `path`
(`mnist/*-idx?-ubyte` as fetched by data/fetch_mnist.sh, `cifar-10-batches-py/` as fetched by
data/fetch_cifar10.sh), same return conventions (raw intensities in [0, 255] as float64, zero-based integer
labels); nothing is downloaded."""
import os
import pickle
import struct

import numpy as np


def _idx(fname, header_fmt, dtype):
    """one IDX file: big-endian header (magic, dims...) followed by the raw array"""
    with open(fname, 'rb') as f:
        head = struct.unpack(header_fmt, f.read(struct.calcsize(header_fmt)))
        return head[1:], np.frombuffer(f.read(), dtype=dtype)


def load_mnist(mode='train', path='.'):
    """-> data [n, 784] float (0..255), target [n] integer labels"""
    stem = {'train': 'train', 'test': 't10k'}.get(mode)
    if stem is None:
        raise ValueError("`mode` must be 'train' or 'test'")
    d = os.path.join(path, 'mnist')
    (n, rows, cols), pix = _idx(os.path.join(d, stem + '-images-idx3-ubyte'), '>IIII', np.uint8)
    (_n,), lab = _idx(os.path.join(d, stem + '-labels-idx1-ubyte'), '>II', np.int8)
    return pix.reshape(n, rows * cols).astype(float), lab.copy()


def load_cifar10(mode='train', path='.'):
    """-> data [n, 3072] float (0..255, channel-major as stored), target [n] integer labels"""
    files = {'train': ['data_batch_%d' % i for i in range(1, 6)], 'test': ['test_batch']}.get(mode)
    if files is None:
        raise ValueError("`mode` must be 'train' or 'test'")
    data, target = [], []
    for name in files:
        with open(os.path.join(path, 'cifar-10-batches-py', name), 'rb') as f:
            try:
                batch = pickle.load(f)
            except UnicodeDecodeError:          # the python-2 pickles of the original archive
                f.seek(0)
                batch = pickle.load(f, encoding='latin1')
        get = lambda k: batch[k] if k in batch else batch[k.encode()]
        data.append(np.asarray(get('data'), dtype=float))
        target.append(np.asarray(get('labels'), dtype=int))
    return np.concatenate(data), np.concatenate(target)


def im_flatten(X):
    """[n, H, W, 3] (or one [H, W, 3] image) -> [n, 3*H*W] in the channel-major order the models are trained on"""
    X = np.asarray(X)
    single = X.ndim == 3
    X = X[None] if single else X
    out = X.transpose(0, 3, 1, 2).reshape(len(X), -1)
    return out[0] if single else out


def im_unflatten(X):
    """inverse of im_flatten for square images: [n, 3*D*D] (or one vector) -> [n, D, D, 3]"""
    X = np.asarray(X)
    single = X.ndim == 1
    X = X[None] if single else X
    D = int(round(np.sqrt(X.shape[1] // 3)))
    out = X.reshape(len(X), 3, D, D).transpose(0, 2, 3, 1)
    return out[0] if single else out
