"""GPU-side generators for bench.py's synthetic workloads, built with
torch so the BASELINE-sized inputs (100k sketches, 10^10 bases) are produced directly
in HBM.  int64 tensors carry uint64 bit patterns (two's-complement wraparound)."""
import torch

_GOLDEN = -7046029254386353131            # 0x9E3779B97F4A7C15 as int64
_M1 = -4658895280553007687                # 0xBF58476D1CE4E5B9
_M2 = -7723592293110705685                # 0x94D049BB133111EB
_PAD_SORT = (1 << 63) - 1


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def splitmix64(state0, n):
    """state0: int64 tensor [...]; returns outputs 1..n as int64 [..., n]."""
    idx = torch.arange(1, n + 1, device=state0.device, dtype=torch.int64)
    z = state0.unsqueeze(-1) + _GOLDEN * idx
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def clustered_sketch_table(n, s=1000, clusters=1000, seed=0, pool=1500, private=400, keep_p=0.8,
                           length=1_000_000, device="cuda", block=20000, contiguous=False):
    """C3 table on `device`: (hashes int64[n, s] = uint64 bits, padded with -1 (=2^64-1);
    nhash int32[n]; lengths int64[n])."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    hashes = torch.empty((n, s), dtype=torch.int64, device=device)
    nhash = torch.empty(n, dtype=torch.int32, device=device)
    cid = torch.arange(clusters, device=device, dtype=torch.int64)
    pools = _lsr(splitmix64(-3335678366873096957 * (cid + 1), pool), 10)      # 0xD1B54A32D192ED03
    for b0 in range(0, n, block):
        b1 = min(n, b0 + block)
        m = torch.arange(b0, b1, device=device, dtype=torch.int64)
        # SURVEY 8d interleaves the clusters (m % C); contiguous=True lays each cluster out as a
        # run of consecutive rows, the order taxonomically sorted collections have
        cl = (m * clusters) // n if contiguous else m % clusters
        ph = pools[cl]                                                        # [B, pool]
        take = torch.rand((b1 - b0, pool), device=device, generator=gen) < keep_p
        ph = torch.where(take, ph, torch.full_like(ph, _PAD_SORT))
        priv = _lsr(splitmix64(-6882143410218379217 * (m + 1) + seed, private), 10)   # 0xA0761D6478BD642F
        x, _ = torch.sort(torch.cat([ph, priv], dim=1), dim=1)
        dup = torch.zeros_like(x, dtype=torch.bool)
        dup[:, 1:] = x[:, 1:] == x[:, :-1]
        x = torch.where(dup, torch.full_like(x, _PAD_SORT), x)
        x, _ = torch.sort(x, dim=1)
        x = x[:, :s]
        valid = x != _PAD_SORT
        nhash[b0:b1] = valid.sum(dim=1).to(torch.int32)
        hashes[b0:b1] = torch.where(valid, x, torch.full_like(x, -1))
    lengths = torch.full((n,), length, dtype=torch.int64, device=device)
    return hashes, nhash, lengths


def random_sketch_table(n, s=1000, seed=0, bits=54, length=1_000_000, device="cuda", block=20000):
    """SURVEY 8d bracket "all-random": every sketch its own s ascending distinct values below 2^bits
    (common ~ 0 for every pair)."""
    hashes = torch.empty((n, s), dtype=torch.int64, device=device)
    for b0 in range(0, n, block):
        b1 = min(n, b0 + block)
        m = torch.arange(b0, b1, device=device, dtype=torch.int64)
        x = _lsr(splitmix64(-6882143410218379217 * (m + 1) + seed, s + 8), 64 - bits)
        x, _ = torch.sort(x, dim=1)
        dup = torch.zeros_like(x, dtype=torch.bool)
        dup[:, 1:] = x[:, 1:] == x[:, :-1]
        x = torch.where(dup, torch.full_like(x, _PAD_SORT), x)
        x, _ = torch.sort(x, dim=1)
        assert int((x[:, :s] == _PAD_SORT).sum()) == 0
        hashes[b0:b1] = x[:, :s]
    return hashes, torch.full((n,), s, dtype=torch.int32, device=device), torch.full((n,), length, dtype=torch.int64, device=device)


def identical_sketch_table(n, s=1000, seed=0, length=1_000_000, device="cuda"):
    """SURVEY 8d bracket "all-identical": n copies of one sketch (common = s for every pair)."""
    h, nh, ln = random_sketch_table(1, s, seed=seed, length=length, device=device)
    return h.repeat(n, 1).contiguous(), nh.repeat(n).contiguous(), ln.repeat(n).contiguous()


def clade_sketch_table(n, s=1000, clade=1000, seed=0, device="cuda", contiguous=True):
    """Clades of `clade` near-identical sketches (97 % of a clade's pool of 1.06 s values kept, 2 % private):
    what collections of thousands of isolates of one species look like.  Rows of a clade are consecutive
    (taxonomic order) unless contiguous=False."""
    return clustered_sketch_table(n, s, clusters=max(1, n // clade), seed=seed, pool=int(1.06 * s), private=max(1, s // 50),
                                  keep_p=0.97, device=device, contiguous=contiguous)


def _sp_mix(z):
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def species_sketch_table(n, s=1000, seed=0, q_inner=0.04, q_leaf=0.18, length=1_000_000, device="cuda", block=4096):
    """One species as a tree of descent (see workloads/synth.species_sketches: the same bits): pairs share 10 - 50 % of their
    values, no near-copies, no small common pool; rows in a fixed pseudo-random order."""
    import math
    L = max(1, math.ceil(math.log2(max(n, 2))))
    C1, C2, C3 = _GOLDEN, -4417276706812531889, 1609587929392839161           # 0x9E37.., 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9
    sd = (seed * _GOLDEN) & ((1 << 64) - 1)
    sd = sd - (1 << 64) if sd >= (1 << 63) else sd
    ids = torch.arange(n, device=device, dtype=torch.int64)
    order = torch.argsort(_sp_mix(ids * C3 ^ sd).to(torch.float64) + (_sp_mix(ids * C3 ^ sd) < 0) * 18446744073709551616.0, stable=True)
    hashes = torch.empty((n, s), dtype=torch.int64, device=device)
    k = torch.arange(s, device=device, dtype=torch.int64)[None, :]
    for b0 in range(0, n, block):
        b1 = min(n, b0 + block)
        i = order[b0:b1][:, None]                                               # the leaves that land in these rows
        origin = torch.zeros((b1 - b0, s), dtype=torch.int64, device=device)
        for lev in range(1, L + 1):
            a = i >> (L - lev)
            lc = (lev * C2) & ((1 << 64) - 1)
            lc = lc - (1 << 64) if lc >= (1 << 63) else lc                      # (wrapped to int64 like the tensors' products)
            u = _sp_mix((k * C1) ^ lc ^ (a * C3) ^ sd)
            thr = int((q_leaf if lev == L else q_inner) * 4294967296.0)
            rep = _lsr(u, 32) < thr
            origin = torch.where(rep, (lev << 40) | a, origin)
        hashes[b0:b1] = (k << 44) + (_sp_mix((k * C2) ^ (origin * C1) ^ sd) & ((1 << 44) - 1))
    return hashes, torch.full((n,), s, dtype=torch.int32, device=device), torch.full((n,), length, dtype=torch.int64, device=device)


def synthetic_genomes(g_begin, g_end, length, device="cuda", block=256, stride=1):
    """ASCII bases uint8[(g_end-g_begin), length] of synthetic genomes g_begin..g_end-1
    (same definition as workloads.synth.synthetic_genome).

    SURVEY.md section 8d seeds genome g with GOLDEN*(g+1), and GOLDEN is also splitmix64's
    state increment: consecutive genomes are the same stream shifted by one word (32 bases).
    That is harmless for sketch throughput (the work per k-mer does not depend on the data)
    but makes every genome contain every other's k-mers; workloads that need unrelated
    genomes (screen) pass stride > length/32 to take genome ids g*stride, whose streams
    do not overlap."""
    n = g_end - g_begin
    out = torch.empty((n, length), dtype=torch.uint8, device=device)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    nw = (length + 31) // 32
    shifts = torch.arange(32, device=device, dtype=torch.int64) * 2
    for b0 in range(0, n, block):
        b1 = min(n, b0 + block)
        g = torch.arange(g_begin + b0, g_begin + b1, device=device, dtype=torch.int64) * stride
        w = splitmix64(_GOLDEN * (g + 1), nw)                                 # [B, nw]
        codes = ((w.unsqueeze(-1) >> shifts) & 3).reshape(b1 - b0, nw * 32)[:, :length]
        out[b0:b1] = lut[codes]
    return out


def synthetic_reads(genomes, n_reads, read_len=150, seed=0, err=0.005):
    """uint8[n_reads, read_len + 1] on the GPU: reads sampled uniformly from both strands of
    `genomes` (uint8[G, L] ASCII ACGT) with substitution errors at rate `err`, each followed
    by the record separator (0x0A) -- the buffer can be handed to mg_screen_add_dev /
    mg_sketch_dev as is."""
    dev = genomes.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    ng, length = genomes.shape
    gi = torch.randint(0, ng, (n_reads,), device=dev, generator=gen)
    st = torch.randint(0, length - read_len, (n_reads,), device=dev, generator=gen)
    idx = (gi * length + st).unsqueeze(1) + torch.arange(read_len, device=dev).unsqueeze(0)
    r = genomes.reshape(-1)[idx]
    code = ((r >> 1) & 3).to(torch.int64)                 # ASCII bits 1-2: A=0 C=1 T=2 G=3
    hit = torch.rand((n_reads, read_len), device=dev, generator=gen) < err
    shift = torch.randint(1, 4, (n_reads, read_len), device=dev, generator=gen)
    code = torch.where(hit, (code + shift) & 3, code)     # substitute by one of the other three bases
    rc = torch.rand((n_reads,), device=dev, generator=gen) < 0.5
    code = torch.where(rc.unsqueeze(1), torch.flip(code ^ 2, dims=[1]), code)   # complement = code ^ 2
    lut = torch.tensor([65, 67, 84, 71], dtype=torch.uint8, device=dev)
    out = torch.full((n_reads, read_len + 1), 10, dtype=torch.uint8, device=dev)
    out[:, :read_len] = lut[code]
    return out

