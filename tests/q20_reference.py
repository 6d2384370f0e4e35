"""NumPy statement of the packed factor image of csrc/foldq.hip ("Q20"): test infrastructure — the encoder restated
row by row and the decoder that defines what the bits mean.  The GPU tests decode the DEVICE's image with `decode` (bit
exact against pk_q20_decode_f64) and hold the device encoder to the format's contract (every row within its own D_j);
the CPU tests run the restated encoder against the same contract."""
import numpy as np

TAB = 96


def lanes(K):
    for L, cap in ((2, 12), (4, 25), (8, 50), (16, 101), (32, 202)):
        if K <= cap:
            return L
    return 0


def steps(L):
    """entries a lane multiplies in fp32 before its partial sums go to fp64 (csrc/foldq.hip: q20_steps)"""
    return 4 if L >= 8 else (2 if L == 4 else 1)


def kappa(K):
    return np.sqrt(float(K) + 32.0) * (2112.0 + (steps(lanes(K)) + 2.0) * 128.0) * 1.03 / 126.0


def f32(x):
    """fp32 rounding of int32 windows held in float64 / int64 arrays (what v_cvt_f32_i32 does), back in float64"""
    return np.asarray(x, dtype=np.int64).astype(np.int32).astype(np.float32).astype(np.float64)


def bracket(j):
    j = np.asarray(j, dtype=np.int64)
    e = np.floor(np.log2(np.maximum(j, 1))).astype(np.int64)
    b = 4 * (e - 1) + ((j >> np.maximum(e - 2, 0)) & 3)
    return np.where(j < 4, j, b)


def scales(V):
    n = V.shape[0]
    m = np.zeros(TAB)
    np.maximum.at(m, bracket(np.arange(n)), np.abs(V).max(axis=1))
    return m * (1.0 + 2.0 ** -20) / (2.0 ** 31 - 8192.0)


def _windows(d):
    """d: uint32 [n, L, 4] -> int64 windows [n, L, 7] (v0..v5, digit) as the kernels read them"""
    d = d.astype(np.uint64)
    d0, d1, d2, d3 = d[..., 0], d[..., 1], d[..., 2], d[..., 3]

    def align(hi, lo, sh):
        return ((((hi << np.uint64(32)) | lo) >> np.uint64(sh)) & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)
    w = [align(d0, d1, 24), align(d0, d1, 4), align(d1, d2, 16), align(d2, d3, 28), align(d2, d3, 8),
         ((d3 << np.uint64(12)) & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32),
         d0.astype(np.uint32).view(np.int32)]
    return np.stack(w, axis=-1).astype(np.int64)


def decode(img, tab, K):
    """img: uint8 [n, L*16] (little-endian dwords) -> fp64 [n, K + 1], column K = D_j"""
    n = img.shape[0]
    L = lanes(K)
    d = np.ascontiguousarray(img).view(np.uint32).reshape(n, L, 4)
    w = f32(_windows(d))
    s = tab[bracket(np.arange(n))]
    out = np.zeros((n, K + 1))
    for l in range(L):
        for t in range(6):
            c = 6 * l + t
            if c < K:
                out[:, c] = w[:, l, t] * s
    a6 = w[:, :, 6] * s[:, None]
    for e in range((L - 1) // 3):
        c = 6 * L + e
        if c < K and 3 * e + 2 <= L - 2:
            out[:, c] = 2.0 * (a6[:, 3 * e] + a6[:, 3 * e + 1] / 256.0 + a6[:, 3 * e + 2] / 65536.0)
    out[:, K] = kappa(K) * a6[:, L - 1]
    return out


def encode(V):
    """(img uint8 [n, L*16], tab fp64 [96]) by the algorithm of q20_encode_kernel"""
    V = np.asarray(V, dtype=np.float64)
    n, K = V.shape
    L = lanes(K)
    tab = scales(V)
    s = tab[bracket(np.arange(n))]
    inv = np.where(s > 0, 1.0 / np.where(s > 0, s, 1.0), 0.0)
    p = np.zeros((n, L, 6), dtype=np.int64)
    sq = np.zeros(n)           # squared quantisation error (real units)
    mag = np.zeros(n)          # squared magnitude of what the kernel multiplies (window units)
    for l in range(L):
        g = np.zeros(n)
        for t in range(5, -1, -1):
            c = 6 * l + t
            v = V[:, c] if c < K else np.zeros(n)
            f = np.clip(np.rint((v * inv - g) / 4096.0), -524288.0, 524287.0)
            window = f32(f * 4096.0 + g)
            if c < K:
                sq += (v - window * s) ** 2
                mag += window ** 2
            p[:, l, t] = f.astype(np.int64) & 0xFFFFF
            g = (p[:, l, t] >> 8).astype(np.float64)
    G = ((p[:, :, 0] << 4) | (p[:, :, 1] >> 16)).astype(np.float64)
    dig = np.zeros((n, L))
    for e in range((L - 1) // 3):
        c = 6 * L + e
        if not (c < K and 3 * e + 2 <= L - 2):
            continue
        xv = V[:, c]
        X = xv * inv * 0.5
        G0, G1, G2 = G[:, 3 * e], G[:, 3 * e + 1], G[:, 3 * e + 2]
        # every digit is chosen so that the remainder lies in the range the lower digits can represent
        lo2 = (-2147483648.0 + G2) / 65536.0
        lo1 = (-2147483648.0 + G1) / 256.0 + lo2
        d0 = np.clip(np.floor((X - G0 - lo1) / 16777216.0), -128.0, 127.0)
        w0 = f32(d0 * 16777216.0 + G0)
        R0 = X - w0
        d1 = np.clip(np.floor(((R0 - lo2) * 256.0 - G1) / 16777216.0), -128.0, 127.0)
        w1 = f32(d1 * 16777216.0 + G1)
        R1 = R0 - w1 / 256.0
        d2 = np.clip(np.rint((R1 * 65536.0 - G2) / 16777216.0), -128.0, 127.0)
        w2 = f32(d2 * 16777216.0 + G2)
        rep = w0 + w1 / 256.0 + w2 / 65536.0
        dig[:, 3 * e], dig[:, 3 * e + 1], dig[:, 3 * e + 2] = d0, d1, d2
        sq += (xv - 2.0 * rep * s) ** 2
        mag += (2.0 * (np.abs(w0) + np.abs(w1) / 256.0 + np.abs(w2) / 65536.0)) ** 2
    D = (np.sqrt(sq) * 16777216.0 + (steps(L) + 2.0) * np.sqrt(mag) * s) * (1.0 + 2.0 ** -9)
    Y = np.where(s > 0, D / np.where(s > 0, s * kappa(K), 1.0), 0.0)
    GL = G[:, L - 1]
    dw = np.where(s > 0, np.ceil((Y - GL) / 16777216.0), 0.0)
    dw = np.where((s > 0) & (dw <= 127.0) & (f32(np.clip(dw, -128, 127) * 16777216.0 + GL) < Y), dw + 1.0, dw)
    assert dw.max() <= 127.0, 'kappa too small'
    dig[:, L - 1] = np.maximum(dw, -128.0)
    db = dig.astype(np.int64) & 0xFF
    d = np.zeros((n, L, 4), dtype=np.uint32)
    d[..., 0] = ((db << 24) | (p[:, :, 0] << 4) | (p[:, :, 1] >> 16)).astype(np.uint32)
    d[..., 1] = (((p[:, :, 1] & 0xFFFF) << 16) | (p[:, :, 2] >> 4)).astype(np.uint32)
    d[..., 2] = (((p[:, :, 2] & 0xF) << 28) | (p[:, :, 3] << 8) | (p[:, :, 4] >> 12)).astype(np.uint32)
    d[..., 3] = (((p[:, :, 4] & 0xFFF) << 20) | p[:, :, 5]).astype(np.uint32)
    return d.reshape(n, L * 4).view(np.uint8).reshape(n, L * 16), tab
