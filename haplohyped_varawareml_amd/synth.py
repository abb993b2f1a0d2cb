"""Deterministic synthetic VCF workloads — bench/test tooling.

The reference ships no generator; the shapes follow BASELINE.json's configs:
  C2  chr22, 50 000 variants x 1000 samples, biallelic phased          seed 22
  C3  chr1..22, 3 000 000 variants x 2504 samples (GRCh38 proportions) seed 1000 + chrom
  C4  500 000 x 5000 with multiallelic records, missing / half-missing calls, '/' separators and
      GT:DP columns                                                     seed 4
Per-variant table (POS, REF, ALT, allele-frequency threshold) is computed here with numpy; the
genotype bits come from a counter hash (splitmix64 finaliser) so that the HIP renderer
(csrc/synth.hip, writes text straight into HBM) and the numpy renderer below agree byte for byte.
"""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
GOLD = np.uint64(0x9E3779B97F4A7C15)
KV = np.uint64(0xD1B54A32D192ED03)

# GRCh38 autosome lengths (chr1..chr22), used only as shard proportions
GRCH38_LEN = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
              138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
              83257441, 80373285, 58617616, 64444167, 46709983, 50818468]


def mix64(x):
    x = np.asarray(x, dtype=np.uint64).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def _u01(seed, stream, n):
    """n uniforms in [0,1) from the counter hash (stream separates uses)."""
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * GOLD + np.uint64(stream) * KV + np.uint64(seed)
    return (mix64(ctr) >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def _u32(seed, stream, n):
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * GOLD + np.uint64(stream) * KV + np.uint64(seed)
    return (mix64(ctr) >> np.uint64(32)).astype(np.uint32)


def shard_sizes(total_variants=3_000_000):
    """V_c proportional to GRCh38 autosome lengths, sum == total_variants."""
    tot = float(sum(GRCH38_LEN))
    v = [int(round(total_variants * l / tot)) for l in GRCH38_LEN]
    v[0] += total_variants - sum(v)
    return v


def variant_table(seed, n_variants, n_samples):
    """POS strictly increasing (gaps ~ geometric, mean 700), REF/ALT distinct in ACGT, allele
    frequency log-uniform on [1/(2S), 0.5] expressed as a uint32 threshold."""
    V, S = int(n_variants), int(n_samples)
    u = _u01(seed, 1, V)
    gaps = 1 + np.floor(-np.log1p(-u) * 700.0).astype(np.int64)
    pos = (10_000 + np.cumsum(gaps)).astype(np.uint32)
    r = _u32(seed, 2, V)
    ref_i = (r & 3).astype(np.int64)
    alt_i = (ref_i + 1 + ((r >> 2) % 3).astype(np.int64)) & 3
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    uf = _u01(seed, 3, V)
    lo = 1.0 / (2.0 * max(S, 1))
    p = lo * np.power(0.5 / lo, uf)
    thr = np.minimum(np.floor(p * 4294967296.0), 4294967295.0).astype(np.uint32)
    return dict(pos=pos, ref=acgt[ref_i].copy(), alt=acgt[alt_i].copy(), thr=thr)


def sample_names(n_samples):
    return [f"S{i + 1:05d}" for i in range(n_samples)]


def header_text(contig, names, contig_length=None):
    h = ["##fileformat=VCFv4.2", '##FILTER=<ID=PASS,Description="All filters passed">']
    h.append(f"##contig=<ID={contig}" + (f",length={contig_length}>" if contig_length else ">"))
    h.append('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">')
    h.append('##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read Depth">')
    h.append("##source=hhgt-synth")
    h.append("\t".join(["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] + list(names)))
    return ("\n".join(h) + "\n").encode()


def ndigits(pos):
    pos = np.asarray(pos, dtype=np.uint64)
    nd = np.ones(pos.shape, dtype=np.int64)
    t = np.uint64(10)
    for _ in range(10):
        nd += (pos >= t).astype(np.int64)
        t = t * np.uint64(10)
    return nd


def fixed_line_lengths(contig, pos, n_samples):
    """bytes per rendered fixed-width line: len(contig) + digits(POS) + 20 + 4*S"""
    return len(contig) + ndigits(pos) + 20 + 4 * int(n_samples)


def genotype_bits(seed, v_first, n_variants, n_samples, thr):
    """alleles[v, s, h] in {0,1} — same rule as csrc/synth.hip."""
    V, S = int(n_variants), int(n_samples)
    key = mix64(np.uint64(seed) + GOLD)
    with np.errstate(over="ignore"):
        kv = key ^ ((np.arange(V, dtype=np.uint64) + np.uint64(v_first)) * KV)
        sh = np.arange(2 * S, dtype=np.uint64) * GOLD
        u = (mix64(kv[:, None] ^ sh[None, :]) >> np.uint64(32)).astype(np.uint32)
    return (u < np.asarray(thr, dtype=np.uint32)[:, None]).astype(np.uint8).reshape(V, S, 2)


def render_fixed_numpy(contig, table, n_samples, seed, v_first=0, with_header=True, names=None):
    """CPU mirror of hhgt_synth_render_fixed (small sizes).  -> (text bytes, line_off uint64[V+1])"""
    S = int(n_samples)
    pos, ref, alt, thr = table["pos"], table["ref"], table["alt"], table["thr"]
    V = len(pos)
    head = header_text(contig, names or sample_names(S)) if with_header else b""
    ll = fixed_line_lengths(contig, pos, S)
    off = np.zeros(V + 1, dtype=np.uint64)
    off[0] = len(head)
    off[1:] = len(head) + np.cumsum(ll).astype(np.uint64)
    out = np.empty(int(off[-1]), dtype=np.uint8)
    out[:len(head)] = np.frombuffer(head, dtype=np.uint8)
    bits = genotype_bits(seed, v_first, V, S, thr)
    cb = contig.encode()
    for v in range(V):
        o = int(off[v])
        pre = cb + b"\t" + str(int(pos[v])).encode() + b"\t.\t" + bytes([ref[v]]) + b"\t" + bytes([alt[v]]) + b"\t.\tPASS\t.\tGT\t"
        out[o:o + len(pre)] = np.frombuffer(pre, dtype=np.uint8)
        o += len(pre)
        seg = out[o:o + 4 * S].reshape(S, 4)
        seg[:, 0] = 48 + bits[v, :, 0]
        seg[:, 1] = ord("|")
        seg[:, 2] = 48 + bits[v, :, 1]
        seg[:, 3] = 9
        seg[S - 1, 3] = 10
    return out.tobytes(), off


def render_mixed(contig, n_variants, n_samples, seed, names=None, p_multi=0.10, p_missing=0.02,
                 p_half=0.005, p_unphased=0.05, p_dp=0.20, p_indel=0.02, crlf=False, trailing_newline=True):
    """C4-style text (pure Python, small sizes): multiallelic records (ALT 'A,C' / 'A,C,G', GT up
    to 3), './.', '.|.', half-missing '.|1' '0|.', '/' separators, records with GT:DP columns, a few
    indels/MNPs and symbolic ALTs that the isSNP filter must drop.  -> bytes"""
    rng = np.random.Generator(np.random.PCG64(seed))
    V, S = int(n_variants), int(n_samples)
    tab = variant_table(seed, V, S)
    eol = "\r\n" if crlf else "\n"
    lines = [header_text(contig, names or sample_names(S)).decode().replace("\n", eol)]
    bases = "ACGT"
    for v in range(V):
        ref = chr(tab["ref"][v])
        alt = chr(tab["alt"][v])
        kind = rng.random()
        n_alt = 1
        if kind < p_multi:
            n_alt = 2 + int(rng.random() < 0.3)
            others = [b for b in bases if b != ref]
            alt = ",".join(others[:n_alt])
        elif kind < p_multi + p_indel:
            c = rng.integers(0, 5)
            if c == 0:
                ref = ref + "T"
            elif c == 1:
                alt = alt + "G"
            elif c == 2:
                alt = "*"
            elif c == 3:
                alt = "<DEL>"
            else:
                alt = alt.lower()
        p = float(tab["thr"][v]) / 4294967296.0
        a = (rng.random((S, 2)) < p).astype(np.int64)
        if n_alt > 1:
            a = a * rng.integers(1, n_alt + 1, size=(S, 2))
        with_dp = rng.random() < p_dp
        miss = rng.random(S)
        unph = rng.random(S) < p_unphased
        dp = rng.integers(0, 100, size=S)
        cols = []
        for s in range(S):
            x, y = str(a[s, 0]), str(a[s, 1])
            if miss[s] < p_missing:
                x = y = "."
            elif miss[s] < p_missing + p_half:
                if rng.random() < 0.5:
                    x = "."
                else:
                    y = "."
            g = x + ("/" if unph[s] else "|") + y
            if with_dp:
                g += ":" + str(int(dp[s]))
            cols.append(g)
        fmt = "GT:DP" if with_dp else "GT"
        lines.append("\t".join([contig, str(int(tab["pos"][v])), ".", ref, alt, ".", "PASS", ".", fmt] + cols) + eol)
    text = "".join(lines)
    if not trailing_newline:
        text = text[: -len(eol)]
    return text.encode()


# ---------------------------------------------------------------------------------------------------
# config-4 style workloads with a device renderer (csrc/synth.hip k_synth_mixed) and this numpy mirror
def mixed_table(seed, n_variants, n_samples, p_multi=0.10, p_weird=0.02, p_dp=0.20):
    """variant_table + REF/ALT strings (<= 8 bytes), n_alt, with_dp, kept (passes isSNP)."""
    V = int(n_variants)
    t = variant_table(seed, V, n_samples)
    u, u2, u3, u5 = _u01(seed, 4, V), _u01(seed, 6, V), _u01(seed, 7, V), _u01(seed, 5, V)
    ref_s = np.zeros((V, 8), np.uint8)
    alt_s = np.zeros((V, 8), np.uint8)
    ref_len = np.ones(V, np.uint32)
    alt_len = np.ones(V, np.uint32)
    n_alt = np.ones(V, np.uint32)
    ref_s[:, 0] = t["ref"]
    alt_s[:, 0] = t["alt"]
    acgt = b"ACGT"
    multi = u < p_multi
    weird = (~multi) & (u < p_multi + p_weird)
    for v in np.nonzero(multi)[0]:
        k = 2 + int(u2[v] < 0.3)
        others = [b for b in acgt if b != t["ref"][v]][:k]
        sx = b",".join(bytes([b]) for b in others)
        alt_s[v, :len(sx)] = np.frombuffer(sx, np.uint8)
        alt_len[v] = len(sx)
        n_alt[v] = k
    for v in np.nonzero(weird)[0]:
        c = int(u3[v] * 5)
        if c == 0:
            ref_s[v, 1] = ord("T"); ref_len[v] = 2
        elif c == 1:
            alt_s[v, 1] = ord("G"); alt_len[v] = 2
        elif c == 2:
            alt_s[v, 0] = ord("*")
        elif c == 3:
            alt_s[v, :5] = np.frombuffer(b"<DEL>", np.uint8); alt_len[v] = 5
        else:
            alt_s[v, 0] = t["alt"][v] + 32          # lower case
    with_dp = (u5 < p_dp)
    t.update(ref_s=ref_s, alt_s=alt_s, ref_len=ref_len, alt_len=alt_len, n_alt=n_alt, with_dp=with_dp,
             kept=(~multi) & (~weird),
             meta=(ref_len | (alt_len << 4) | (n_alt << 8) | (with_dp.astype(np.uint32) << 12)).astype(np.uint32),
             ref8=np.ascontiguousarray(ref_s).view("<u8").reshape(-1), alt8=np.ascontiguousarray(alt_s).view("<u8").reshape(-1))
    return t


def mixed_line_lengths(contig, t, n_samples):
    fmt_len = np.where(t["with_dp"], 5, 2)
    fw = np.where(t["with_dp"], 7, 4)
    return (len(contig) + ndigits(t["pos"]) + t["ref_len"].astype(np.int64) + t["alt_len"].astype(np.int64) + 16
            + fmt_len + int(n_samples) * fw)


def mixed_calls(seed, t, n_samples, v_idx, v_first=0):
    """per-call text pieces for variants v_idx: (c0, sep, c1, dp) uint8 arrays [len(v_idx), S]"""
    S = int(n_samples)
    v_idx = np.asarray(v_idx, dtype=np.uint64)
    key = mix64(np.uint64(seed) + GOLD)
    A5 = np.uint64(0xA5A5A5A5A5A5A5A5)
    with np.errstate(over="ignore"):
        kv = key ^ ((v_idx + np.uint64(v_first)) * KV)
        s2 = np.arange(S, dtype=np.uint64) * np.uint64(2)
        r0 = mix64(kv[:, None] ^ (s2 * GOLD)[None, :])
        r1 = mix64(kv[:, None] ^ ((s2 + np.uint64(1)) * GOLD)[None, :])
        r2 = mix64(kv[:, None] ^ (s2 * GOLD)[None, :] ^ A5)
    thr = t["thr"][v_idx.astype(np.int64)].astype(np.uint64)[:, None]
    na = t["n_alt"][v_idx.astype(np.int64)].astype(np.uint64)[:, None]

    def allele(r):
        hit = (r >> np.uint64(32)) < thr
        multi = np.uint64(1) + ((r >> np.uint64(8)) & np.uint64(0xFFFFFF)) % np.maximum(na, np.uint64(1))
        return np.where(hit, np.where(na > 1, multi, np.uint64(1)), np.uint64(0)).astype(np.int64)

    a0, a1 = allele(r0), allele(r1)
    mm = (r2 & np.uint64(0xFFFF)).astype(np.int64)
    both = mm < 1311
    half = (~both) & (mm < 1639)
    which1 = ((r2 >> np.uint64(16)) & np.uint64(1)).astype(bool)
    miss0 = both | (half & ~which1)
    miss1 = both | (half & which1)
    c0 = np.where(miss0, ord("."), 48 + a0).astype(np.uint8)
    c1 = np.where(miss1, ord("."), 48 + a1).astype(np.uint8)
    sep = np.where(((r2 >> np.uint64(20)) & np.uint64(0xFFFF)).astype(np.int64) < 3277, ord("/"), ord("|")).astype(np.uint8)
    dp = (10 + ((r2 >> np.uint64(40)) % np.uint64(90)).astype(np.int64)).astype(np.uint8)
    g0 = np.where(miss0, -9, a0).astype(np.int8)
    g1 = np.where(miss1, -9, a1).astype(np.int8)
    return c0, sep, c1, dp, np.stack([g0, g1], axis=-1)


def mixed_expected_G(seed, t, n_samples, v_idx, v_first=0):
    """int8 [S, len(v_idx), 2] the encode path must produce for the KEPT variants v_idx"""
    return mixed_calls(seed, t, n_samples, v_idx, v_first)[4].transpose(1, 0, 2).copy()


def render_mixed_numpy(contig, t, n_samples, seed, v_first=0, with_header=True, names=None):
    """CPU mirror of hhgt_synth_render_mixed (small sizes) -> (bytes, line_off)"""
    S = int(n_samples)
    V = len(t["pos"])
    head = header_text(contig, names or sample_names(S)) if with_header else b""
    ll = mixed_line_lengths(contig, t, S)
    off = np.zeros(V + 1, dtype=np.uint64)
    off[0] = len(head)
    off[1:] = len(head) + np.cumsum(ll).astype(np.uint64)
    out = bytearray(int(off[-1]))
    out[:len(head)] = head
    c0, sep, c1, dp, _ = mixed_calls(seed, t, S, np.arange(V), v_first)
    cb = contig.encode()
    for v in range(V):
        ref = bytes(t["ref_s"][v, :t["ref_len"][v]])
        alt = bytes(t["alt_s"][v, :t["alt_len"][v]])
        fmt = b"GT:DP" if t["with_dp"][v] else b"GT"
        cols = []
        for s in range(S):
            f = bytes([c0[v, s], sep[v, s], c1[v, s]])
            if t["with_dp"][v]:
                f += b":%02d" % int(dp[v, s])
            cols.append(f)
        line = b"\t".join([cb, str(int(t["pos"][v])).encode(), b".", ref, alt, b".", b"PASS", b".", fmt] + cols) + b"\n"
        o = int(off[v])
        assert len(line) == int(off[v + 1]) - o, (v, len(line), int(off[v + 1]) - o)
        out[o:o + len(line)] = line
    return bytes(out), off
