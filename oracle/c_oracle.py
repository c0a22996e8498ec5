"""ctypes front end of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
        _lib.or_counter_normal.restype = C.c_float
        _lib.or_counter_normal.argtypes = [C.c_uint64, C.c_uint64]
        _lib.or_bpr_step.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32) if a is not None else None


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def init_normal(count, seed, stddev):
    w = np.empty(count, np.float32)
    lib().or_init_normal(_p(w), C.c_int64(count), C.c_uint64(seed), C.c_float(stddev))
    return w


def init_rows(rows, k, seed, stddev):
    """Rows `rows` of a [*, k] table initialised by dr_init_normal_f32(seed, stddev) (no table is built)."""
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    w = np.empty((len(rows), k), np.float32)
    lib().or_init_rows(_p(w), _p(rows), C.c_int64(len(rows)), C.c_int(k), C.c_uint64(seed), C.c_float(stddev))
    return w


def sample_negatives(train_ptr, train_idx, users, num_item, seed, offset=0, n_neg=1, first_item=0):
    """or_sample_negatives: uniform draws from all items minus the user's train items (bprmf.py:126-133,175-181)."""
    train_ptr, train_idx, users = _i32(train_ptr), _i32(train_idx), _i32(users)
    out = np.empty((len(users), n_neg), np.int32)
    lib().or_sample_negatives(_p(train_ptr), _p(train_idx), _p(users), C.c_int64(len(users)), C.c_int32(n_neg),
                              C.c_int64(len(train_ptr) - 1), C.c_int64(num_item), C.c_int64(first_item),
                              C.c_uint64(seed), C.c_uint64(offset), _p(out))
    return out


def sample_negatives_excl(targets, num_item, seed, offset=0, n_neg=1, first_item=1):
    """or_sample_negatives_excl: uniform draws from all items minus the row's own targets (AttRec.py:448-466)."""
    targets = np.ascontiguousarray(targets, dtype=np.int32)
    n, T = targets.shape
    out = np.empty((n, n_neg), np.int32)
    lib().or_sample_negatives_excl(_p(targets), C.c_int64(n), C.c_int32(T), C.c_int32(n_neg), C.c_int64(num_item),
                                   C.c_int64(first_item), C.c_uint64(seed), C.c_uint64(offset), _p(out))
    return out


def pair_dot(A, B, ai, bi):
    A, B, ai, bi = _f32(A), _f32(B), _i32(ai), _i32(bi)
    out = np.empty(len(ai), np.float32)
    lib().or_pair_dot(_p(A), _p(B), C.c_int(A.shape[1]), _p(ai), _p(bi), C.c_int64(len(ai)), _p(out))
    return out


def score_matrix(U, I, kind=0, U2=None, I2=None, alpha=0.0, bias=None):
    U, I, U2, I2, bias = _f32(U), _f32(I), _f32(U2), _f32(I2), _f32(bias)
    out = np.empty((U.shape[0], I.shape[0]), np.float32)
    lib().or_score_matrix(_p(U), _p(U2), C.c_int64(U.shape[0]), _p(I), _p(I2), _p(bias), C.c_int64(I.shape[0]),
                          C.c_int(U.shape[1]), C.c_int(kind), C.c_float(alpha), _p(out))
    return out


def gather(table, ids):
    table, ids = _f32(table), _i32(ids)
    out = np.empty((len(ids), table.shape[1]), np.float32)
    lib().or_gather(_p(table), C.c_int64(table.shape[0]), C.c_int(table.shape[1]), _p(ids), C.c_int64(len(ids)), _p(out))
    return out


def bpr_step(P, Q, slotsP, slotsQ, u, i, j, adam, lr=0.0, alpha=0.0, beta1=0.9, beta2=0.999, eps=1e-8, reg=0.0):
    """In place on the given fp32 arrays; returns the loss sum."""
    u, i, j = _i32(u), _i32(i), _i32(j)
    Pm, Pv = (slotsP + [None, None])[:2] if adam else (None, None)
    Qm, Qv = (slotsQ + [None, None])[:2] if adam else (None, None)
    return lib().or_bpr_step(_p(P), _p(Pm), _p(Pv), _p(Q), _p(Qm), _p(Qv), C.c_int(P.shape[1]), _p(u), _p(i), _p(j),
                             C.c_int64(len(u)), C.c_int(1 if adam else 0), C.c_float(lr), C.c_float(alpha),
                             C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(reg))
