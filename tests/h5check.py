"""Independent look at an HDF5 file through the image's libhdf5 1.10.6 (h5py 3.3 under /opt/conda), run as a
subprocess of that interpreter by tests/test_h5file.py and the converter tests:
    /opt/conda/bin/python3.9 tests/h5check.py FILE OUT.npz [dataset-with-chunks-to-dump ...]
Writes every dataset it can read (unfiltered ones) into OUT.npz, plus for each named chunked dataset its creation
properties and all raw chunks (read_direct_chunk), keyed '<name>|meta' (JSON) and '<name>|chunk|<offsets>'."""
import json
import sys

import h5py
import numpy as np

path, out = sys.argv[1], sys.argv[2]
dump = set(sys.argv[3:])
res = {}
with h5py.File(path, "r") as f:
    names = []

    def walk(g, prefix):          # by link name (visititems reports a hard-linked object once)
        for k in g.keys():
            o = g[k]
            names.append((prefix + k, isinstance(o, h5py.Dataset)))
            if isinstance(o, h5py.Group):
                walk(o, prefix + k + "/")

    walk(f, "")
    res["|names"] = np.array(json.dumps(names))
    for n, is_ds in names:
        if not is_ds:
            continue
        d = f[n]
        plist = d.id.get_create_plist()
        filters = [plist.get_filter(i) for i in range(plist.get_nfilters())]
        meta = dict(shape=list(d.shape), dtype=str(d.dtype), chunks=list(d.chunks) if d.chunks else None,
                    filters=[[int(fl[0]), int(fl[1]), [int(v) for v in fl[2]], fl[3].decode() if isinstance(fl[3], bytes) else str(fl[3])]
                             for fl in filters])
        res[n + "|meta"] = np.array(json.dumps(meta))
        if not filters:
            a = d[...]
            if a.dtype.names:      # compound: one entry per member (structured npz entries do not survive numpy versions)
                meta["fields"] = [[k, str(a.dtype.fields[k][0]), int(a.dtype.fields[k][1])] for k in a.dtype.names]
                meta["itemsize"] = int(a.dtype.itemsize)
                res[n + "|meta"] = np.array(json.dumps(meta))
                for k in a.dtype.names:
                    res[n + "|field|" + k] = np.ascontiguousarray(a[k])
            else:
                res[n] = a
        if n in dump:
            nchunks = d.id.get_num_chunks()
            meta["n_chunks"] = int(nchunks)
            res[n + "|meta"] = np.array(json.dumps(meta))
            for i in range(nchunks):
                info = d.id.get_chunk_info(i)
                mask, raw = d.id.read_direct_chunk(info.chunk_offset)
                assert len(raw) == info.size
                res[n + "|chunk|" + ",".join(str(int(o)) for o in info.chunk_offset)] = np.frombuffer(raw, np.uint8)
                res[n + "|mask|" + ",".join(str(int(o)) for o in info.chunk_offset)] = np.array(int(mask))
np.savez(out, **res)
print("ok", len(res))
