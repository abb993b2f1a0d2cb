"""`fasta_encoder` — the reference's FASTA -> one-hot -> Blosc2 store CLI on the MI355X path.

Same surface as /root/reference/src/haplohyped/fasta_encoder.py: click options --fasta --outdir --cores
(:189-193), class ReferenceGenome(fasta_file, encode_spec, hdf5_file, output_dir) with parse_encode_list,
encode_sequence, load_chromosome, load_genome_parallel, get_sequence; output under OUT/.  Encoding rule
(:47-78): upper-case, every base outside A,C,G,T becomes N, one uint8 column per base with the columns in
SORTED order (A,C,G,N,T for the default spec; `df_onehot.select(sorted(columns))`, :60).
Different inside: the one-hot rows are produced by hhgt_onehot_bases_u8 and compressed by
hhgt_compress_chunks (the same codec the reference reaches through create_dataset(..., compression=32001),
:91,134); the container is the chunk store of store.py (h5py/hdf5plugin are not available here), group
`<chrom>` = the reference's dataset `<chrom>/sequence`.
"""
import ctypes as C
import json
import logging
import os

import click
import numpy as np

logger = logging.getLogger("haplohyped.fasta_encoder")

CHUNK_ROWS = 1 << 18   # rows (bases) per stored chunk


class ReferenceGenome:
    def __init__(self, fasta_file=None, encode_spec=None, hdf5_file=None, output_dir=None, ctx=None):
        self.encode_spec = self.parse_encode_list(encode_spec)
        self.output_dir = output_dir
        self.fasta_file = fasta_file
        self.hdf5_file = hdf5_file
        self.genome_df = None
        self._ctx = ctx
        self._contigs = None

    @staticmethod
    def parse_encode_list(encode_spec):
        """list of single-base byte strings, with the behaviour of fasta_encoder.py:32-45: nothing given = A,C,G,T,N;
        a string is split into its characters; list / tuple items may be str or bytes; anything else is a TypeError
        with the reference's message"""
        spec = encode_spec or "ACGTN"
        if not isinstance(spec, (str, list, tuple)):
            raise TypeError("Please input string or list of strings!")
        return [b.encode() if isinstance(b, str) else b for b in spec]

    # ---- helpers -----------------------------------------------------------------------------------
    def _context(self):
        if self._ctx is None:
            from .device import Context
            self._ctx = Context(0)
        return self._ctx

    def columns(self):
        """sorted column order of the reference's to_dummies frame (:55-60)"""
        return sorted(b.decode() for b in self.encode_spec)

    def _lut(self):
        cols = self.columns()
        lut = np.full(256, cols.index("N") if "N" in cols else 255, np.uint8)
        for b in "ACGT":
            ch = cols.index(b) if b in cols else 255
            lut[ord(b)] = ch
            lut[ord(b.lower())] = ch          # ignore_case=True (:63-78)
        return lut, len(cols)

    def encode_sequence(self, seq_data, ignore_case=True):
        """-> uint8 [L, C] numpy (device kernel behind it)"""
        import torch
        from . import _lib
        if isinstance(seq_data, str):
            arr = np.frombuffer(seq_data.encode(), dtype=np.uint8)
        elif isinstance(seq_data, np.ndarray):
            arr = np.ascontiguousarray(seq_data).view(np.uint8).reshape(-1)
        else:
            raise TypeError("Please input as string or numpy array!")
        ctx = self._context()
        lut, nch = self._lut()
        if not ignore_case:
            for b in "acgt":
                lut[ord(b)] = lut[ord("N")] if "N" in self.columns() else 255
        d = torch.from_numpy(arr.copy()).to(ctx.device)
        out = torch.empty((arr.size, nch), dtype=torch.uint8, device=ctx.device)
        with torch.cuda.device(ctx.device):
            _lib.check(ctx.lib.hhgt_onehot_bases_u8(ctx.h, C.c_void_p(d.data_ptr()), arr.size, lut.ctypes.data, nch,
                                                   C.c_void_p(out.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out.cpu().numpy()

    def _fasta(self):
        if self._contigs is None:
            from .dataset import read_fasta
            self._contigs = read_fasta(self.fasta_file)
        return self._contigs

    def load_chromosome(self, chrom):
        """one contig -> one-hot rows -> framed chunks on disk (fasta_encoder.py:80-96)"""
        import torch
        from . import _lib
        logger.info(f"Encoding chromosome {chrom} from FASTA file {self.fasta_file}")
        bases = self._fasta()[chrom]
        ctx = self._context()
        lut, nch = self._lut()
        L = bases.size
        rows_pad = -(-max(L, 1) // CHUNK_ROWS) * CHUNK_ROWS
        d = torch.from_numpy(bases).to(ctx.device)
        out = torch.zeros(rows_pad * nch, dtype=torch.uint8, device=ctx.device)
        with torch.cuda.device(ctx.device):
            _lib.check(ctx.lib.hhgt_onehot_bases_u8(ctx.h, C.c_void_p(d.data_ptr()), L, lut.ctypes.data, nch,
                                                   C.c_void_p(out.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        chunk_nbytes = CHUNK_ROWS * nch
        from .device import BLOSC1      # filter 32001 = hdf5-blosc: Blosc-1 chunk framing
        dst, off, total = ctx.compress(out, chunk_nbytes, typesize=1, blocksize=32768, fmt=BLOSC1)
        gdir = os.path.join(self.output_dir, chrom)
        os.makedirs(gdir, exist_ok=True)
        dst[:total].cpu().numpy().tofile(os.path.join(gdir, "chunks.bin"))
        np.save(os.path.join(gdir, "offsets.npy"), off.cpu().numpy().astype(np.uint64))
        json.dump(dict(length=int(L), columns=self.columns(), chunk_rows=CHUNK_ROWS, typesize=1, blocksize=32768,
                       raw_bytes=int(L) * nch, compressed_bytes=int(total)), open(os.path.join(gdir, "meta.json"), "w"))
        return chrom, gdir

    def load_genome_parallel(self):
        """chr1..chr22 when present (fasta_encoder.py:98-109); contigs are independent, the device pass is serial"""
        chrom_list = [f"chr{i}" for i in range(1, 23)]
        have = self._fasta()
        results = [self.load_chromosome(c) for c in chrom_list if c in have]
        self.genome_df = dict(results)
        return self.genome_df

    def get_sequence(self, chrom, start, end):
        """int8 [end-start, C] one-hot rows (fasta_encoder.py:111-118), decoded on the device"""
        return self.get_sequence_device(chrom, start, end).cpu().numpy()

    def _h5(self):
        """`hdf5_file` may be the store directory or the single file HDF5Handler.save_to_hdf5 wrote
        (OUT/reference_genome.h5, what the reference's ReferenceGenome(hdf5_file=...) is given)"""
        if self.hdf5_file and os.path.isfile(self.hdf5_file):
            if getattr(self, "_h5r", None) is None:
                from .h5file import H5Reader
                self._h5r = H5Reader(self.hdf5_file)
                self._h5d = {}
            return self._h5r
        return None

    def contigs(self):
        r = self._h5()
        if r is not None:
            return sorted(r.group())
        return sorted(json.load(open(os.path.join(self.hdf5_file or self.output_dir, "meta.json")))["contigs"])

    def contig_meta(self, chrom):
        r = self._h5()
        if r is not None:
            if chrom not in self._h5d:
                info = r.dataset(f"{chrom}/sequence")
                first = r.read_chunk(info, (0, 0))
                self._h5d[chrom] = dict(info=info, length=int(info["shape"][0]), chunk_rows=int(info["chunk_shape"][0]),
                                        columns=[c.decode() for c in r.read_array(f"{chrom}/columns")], typesize=1,
                                        blocksize=int(first[8:12].view("<u4")[0]))
            return self._h5d[chrom]
        gdir = self.genome_df[chrom] if self.genome_df else os.path.join(self.hdf5_file or self.output_dir, chrom)
        return json.load(open(os.path.join(gdir, "meta.json")))

    def get_sequence_device(self, chrom, start, end):
        import torch
        meta = self.contig_meta(chrom)
        nch, rows = len(meta["columns"]), meta["chunk_rows"]
        start, end = max(0, int(start)), min(int(end), meta["length"])
        if end <= start:
            return torch.zeros((0, nch), dtype=torch.int8, device=self._context().device)
        c0, c1 = start // rows, (end - 1) // rows + 1
        ctx = self._context()
        if self._h5() is not None:
            parts = [self._h5r.read_chunk(meta["info"], (i * rows, 0)) for i in range(c0, c1)]
            off = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.uint64)
            src = torch.from_numpy(np.concatenate(parts)).to(ctx.device)
            d_off = torch.from_numpy(off.astype(np.int64)).to(ctx.device)
        else:
            gdir = self.genome_df[chrom] if self.genome_df else os.path.join(self.hdf5_file or self.output_dir, chrom)
            off = np.load(os.path.join(gdir, "offsets.npy"))
            mm = np.memmap(os.path.join(gdir, "chunks.bin"), dtype=np.uint8, mode="r")
            src = torch.from_numpy(np.array(mm[int(off[c0]):int(off[c1])])).to(ctx.device)
            d_off = torch.from_numpy((off[c0:c1 + 1] - off[c0]).astype(np.int64)).to(ctx.device)
        out, bad = ctx.decompress(src, d_off, c1 - c0, rows * nch, typesize=1, blocksize=meta["blocksize"])
        if bad:
            raise RuntimeError(f"{bad} corrupt chunk(s) of {chrom} in {self.hdf5_file or self.output_dir}")
        return out.view(torch.int8).view(-1, nch)[start - c0 * rows: end - c0 * rows]


class HDF5Handler:
    """the reference's class of the same name (fasta_encoder.py:120-141): all contigs into one HDF5 file,
    `/{chrom}/sequence` behind filter 32001 — here laid out natively (h5file.py) from the chunks already on disk"""

    @staticmethod
    def save_to_hdf5(genome_df, hdf5_file):
        from .h5file import FILTER_BLOSC, H5Writer, blosc_cd_values
        logger.info(f"Saving entire reference genome to {hdf5_file}")
        with H5Writer(hdf5_file) as w:
            for chrom, gdir in sorted(dict(genome_df).items()):
                meta = json.load(open(os.path.join(gdir, "meta.json")))
                off = np.load(os.path.join(gdir, "offsets.npy")).astype(np.uint64)
                base = w.append(np.fromfile(os.path.join(gdir, "chunks.bin"), dtype=np.uint8).tobytes())
                nch, rows = len(meta["columns"]), meta["chunk_rows"]
                chunks = [((i * rows, 0), base + int(off[i]), int(off[i + 1] - off[i])) for i in range(len(off) - 1)]
                w.add_chunked(chrom, "sequence", (meta["length"], nch), np.int8, (rows, nch), chunks, filter_id=FILTER_BLOSC,
                              cd_values=blosc_cd_values(1, rows * nch, shuffle=0), filter_name=b"blosc")
                w.add_array(chrom, "columns", np.array([c.encode() for c in meta["columns"]], dtype="S1"))
        logger.info(f"Successfully saved reference genome to {hdf5_file}")
        return hdf5_file


@click.command()
@click.option("--fasta", required=True, type=click.Path(exists=True), help="Path to reference genome FASTA file")
@click.option("--outdir", required=True, type=click.Path(), help="Path to results save folder")
@click.option("--cores", default=os.cpu_count(), type=int, help="Number of CPU cores to use")
def main(fasta, outdir, cores):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    store = os.path.join(outdir, "reference_genome.hhgt")
    os.makedirs(store, exist_ok=True)
    ref_genome = ReferenceGenome(fasta_file=fasta, output_dir=store)
    done = ref_genome.load_genome_parallel()
    json.dump(dict(format="hhgt-reference", version=1, contigs=sorted(done)), open(os.path.join(store, "meta.json"), "w"))
    HDF5Handler.save_to_hdf5(done, os.path.join(outdir, "reference_genome.h5"))      # fasta_encoder.py:195-200
    logger.info(f"Reference genome store created at {store}")


if __name__ == "__main__":
    main()
