/* Test infrastructure: a minimal HDF5 filter plugin for filter id 32001 (Blosc), DECODE side only, restating what
 * hdf5-blosc / hdf5plugin.Blosc do when a chunk is read: ask c-blosc for the sizes in the chunk header, decompress.
 * Built by tests/test_h5file.py against the image's libhdf5 1.10.6 and c-blosc 1.21 (both under /opt/conda) so that
 * libhdf5's own filter pipeline can read the files h5file.py writes — the path a stock h5py + hdf5plugin user takes. */
#include <hdf5.h>
#include <H5PLextern.h>
#include <blosc.h>
#include <stdlib.h>

static size_t blosc_min_filter(unsigned flags, size_t cd_nelmts, const unsigned cd_values[], size_t nbytes, size_t *buf_size,
                               void **buf)
{
    (void)cd_nelmts;
    (void)cd_values;
    (void)nbytes;
    if (!(flags & H5Z_FLAG_REVERSE)) return 0; /* no encoder here: the chunks are produced on the GPU */
    size_t outsize = 0, cbytes = 0, blocksize = 0;
    blosc_cbuffer_sizes(*buf, &outsize, &cbytes, &blocksize);
    if (outsize == 0) return 0;
    void *out = malloc(outsize);
    if (!out) return 0;
    if (blosc_decompress(*buf, out, outsize) <= 0) {
        free(out);
        return 0;
    }
    free(*buf);
    *buf = out;
    *buf_size = outsize;
    return outsize;
}

static const H5Z_class2_t blosc_min_class = {H5Z_CLASS_T_VERS, (H5Z_filter_t)32001, 1, 1, "blosc (minimal test decoder)",
                                             NULL, NULL, (H5Z_func_t)blosc_min_filter};

H5PL_type_t H5PLget_plugin_type(void) { return H5PL_TYPE_FILTER; }
const void *H5PLget_plugin_info(void) { return &blosc_min_class; }
