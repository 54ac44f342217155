#!/usr/bin/env python3
"""Reads the ONE fixture the reference holds for this path -- Test/lunar_lander/params.pickle and opt_state.pickle, written
by General/Base/utils.py:21-29 (the initial hk.Params of LunarLander/dddqn.py:19-22 and the zero optax.adamw state) -- and
writes tests/golden/ref_init_params.npz: raw float32 arrays keyed by the haiku leaf names. DATA ONLY.

The files are NOT unpickled (torch.load(weights_only=True) refuses them: non-allow-listed globals; a pickle load would
import jax / optax / haiku classes and run their reconstructors). `pickletools.genops` only tokenises the byte stream:
nothing is imported, constructed or called. From the token stream this script takes the string keys, the shape tuples and
the raw little-endian byte strings of numpy's ndarray state `(version, shape, dtype, is_fortran, rawdata)`; it checks that
each byte string is exactly prod(shape) * 4 bytes and that the dtype token is `f4` (`i4` for the optax step count).

Run in the build container (the reference does not travel):  python tests/golden/extract_ref_init.py
"""
import os
import pickletools
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/Test/lunar_lander"
MODULES = ("model/~/linear", "model/~/linear_1", "model/~/linear_2", "model/~/linear_3")


sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from deep_q_learning_amd.General.Base.pickle_tokens import read_arrays as arrays_in_order  # noqa: E402  (the token reader)


def main():
    if not os.path.isdir(REF):
        sys.exit("the reference is not present here (this script runs in the build container only)")
    res = {}
    got = arrays_in_order(os.path.join(REF, "params.pickle"))
    assert len(got) == 8, len(got)
    for module, leaf, dt, shape, raw in got:
        assert dt == "<f4" and module in MODULES and leaf in ("w", "b")
        assert len(raw) == 4 * int(np.prod(shape)), (module, leaf, shape, len(raw))
        res[f"{module}/{leaf}"] = np.frombuffer(raw, "<f4").reshape(shape).copy()
    st = arrays_in_order(os.path.join(REF, "opt_state.pickle"))
    # ScaleByAdamState(count, mu, nu) + two EmptyState: one i4 scalar, then 8 mu leaves, then 8 nu leaves
    assert len(st) == 17 and len(st[0][4]) == 4, len(st)
    res["opt/count"] = np.frombuffer(st[0][4], "<i4").copy()
    for which, part in (("mu", st[1:9]), ("nu", st[9:17])):
        for module, leaf, dt, shape, raw in part:
            assert len(raw) == 4 * int(np.prod(shape))
            res[f"opt/{which}/{module}/{leaf}"] = np.frombuffer(raw, "<f4").reshape(shape).copy()
    np.savez_compressed(os.path.join(HERE, "ref_init_params.npz"), **res)
    for k, v in res.items():
        print(f"{k:40s} {str(v.shape):10s} {v.dtype}  |max| {np.abs(v).max():.5f}")


if __name__ == "__main__":
    main()
