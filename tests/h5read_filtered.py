"""Run by /opt/conda's python with HDF5_PLUGIN_PATH pointing at the test-built filter plugin (tests/h5z_blosc_min.c):
reads the named datasets THROUGH libhdf5's filter pipeline and stores them into an npz (compound datasets member-wise).
    python tests/h5read_filtered.py FILE OUT.npz dataset [dataset ...]"""
import sys

import h5py
import numpy as np

path, out = sys.argv[1], sys.argv[2]
res = {}
with h5py.File(path, "r") as f:
    for n in sys.argv[3:]:
        a = f[n][...]
        if a.dtype.names:
            for k in a.dtype.names:
                res[n + "|field|" + k] = np.ascontiguousarray(a[k])
        else:
            res[n] = a
np.savez(out, **res)
print("ok")
