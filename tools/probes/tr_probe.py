"""GPU (tools): semantics and LDS cost of ds_read_b64_tr_b16.  python tools/probes/tr_probe.py > gpurun_out/tr_probe.json
1. semantics: random distinct 8-byte-aligned per-lane addresses; hypothesis (guide, T10): inside every 16-lane group, result lane i,
   element j = element (i & 3) of the 8 bytes addressed by lane 4 j + (i >> 2) of the group.
2. cost: cycles per 16 reads of one wave (4 / 8 waves per workgroup, one workgroup) for candidate layouts of a row-major tile."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "tr_probe.so"))
dev = torch.device("cuda:0")


def sem():
    rng = np.random.default_rng(5)
    ok_all = True
    detail = None
    for trial in range(4):
        slots = rng.permutation(4096)[:64]            # 8-byte slots of the 32 KiB array
        addr = (slots * 8).astype(np.uint32)
        a = torch.from_numpy(addr.view(np.int32)).to(dev)
        out = torch.zeros(256, dtype=torch.int16, device=dev)
        rc = lib.tr_sem(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), None)
        torch.cuda.synchronize()
        assert rc == 0
        got = out.cpu().numpy().view(np.uint16).reshape(64, 4)
        exp = np.zeros((64, 4), dtype=np.uint16)
        for l in range(64):
            g, i = l >> 4, l & 15
            for j in range(4):
                src = g * 16 + 4 * j + (i >> 2)
                exp[l, j] = addr[src] // 2 + (i & 3)
        ok = bool((got == exp).all())
        ok_all &= ok
        if not ok and detail is None:
            # decode: for each result element, which lane's address and which element of it
            m = []
            for l in range(64):
                row = []
                for j in range(4):
                    v = int(got[l, j])
                    src = [(s, v - int(addr[s]) // 2) for s in range(64) if 0 <= v - int(addr[s]) // 2 < 4]
                    row.append(src[0] if src else None)
                m.append(row)
            detail = m[:20]
    return dict(hypothesis_holds=ok_all, decoded_first_lanes=detail)


def table(fn):
    t = np.zeros((16, 64), dtype=np.uint32)
    for k in range(16):
        for l in range(64):
            t[k, l] = fn(k, l)
    return t


def tr128(key):
    def f(k, l):
        df, half, g, i = k >> 1, k & 1, l >> 4, l & 15
        row = 8 * g + 4 * half + (i >> 2)
        c = 2 * df + ((i & 3) >> 1)
        return row * 256 + ((c ^ key(row)) << 4) + (i & 1) * 8
    return f


def b128_first(key):     # A-operand reads of the first product, D = 128: fragment f = k & 1, kd = (k >> 1) & 3 (twice over)
    def f(k, l):
        fr, kd, g, li = k & 1, (k >> 1) & 3, l >> 4, l & 15
        row = 8 * (li >> 2) + 4 * fr + (li & 3)
        return row * 256 + (((kd * 4 + g) ^ key(row)) << 4)
    return f


def tr64(key):
    def f(k, l):
        a, df, half, g, i = k >> 3, (k >> 1) & 3, k & 1, l >> 4, l & 15
        row = 32 * a + 8 * g + 4 * half + (i >> 2)
        c = 2 * df + ((i & 3) >> 1)
        return row * 128 + ((c ^ key(row)) << 4) + (i & 1) * 8
    return f


def b128_first64(key):   # K fragment reads of the D = 64 forward: kf = k & 3, kd = (k >> 2) & 1
    def f(k, l):
        kf, kd, g, li = k & 3, (k >> 2) & 1, l >> 4, l & 15
        row = (kf >> 1) * 32 + (kf & 1) * 4 + (li >> 2) * 8 + (li & 3)
        return row * 128 + (((kd * 4 + g) ^ key(row)) << 4)
    return f


KEYS128 = {
    "none": lambda r: 0,
    "ring(shipped)": lambda r: ((r >> 3) << 2) | (r & 3),
    "new": lambda r: ((r & 3) << 1) | (((r >> 3) & 1) << 3),
}
KEYS64 = {
    "none": lambda r: 0,
    "fwd(shipped)": lambda r: ((((r >> 3) & 3) << 2) | (r & 3)) >> 1,
    "new": lambda r: (((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2),
    "new2": lambda r: (((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2) | (r & 1),
}


def run(kind, tab, waves, iters=2000):
    a = torch.from_numpy(tab.view(np.int32).copy()).to(dev)
    cyc = torch.zeros(waves, dtype=torch.int64, device=dev)
    sink = torch.zeros(64 * waves, dtype=torch.int32, device=dev)
    for _ in range(2):
        rc = lib.tr_time(kind, ctypes.c_void_p(a.data_ptr()), iters, waves, 1, ctypes.c_void_p(cyc.data_ptr()), ctypes.c_void_p(sink.data_ptr()), None)
        torch.cuda.synchronize()
        assert rc == 0
    return float(cyc.float().mean()) / iters


def main():
    res = dict(semantics=sem(), cycles_per_16_reads={})
    cases = {"tr contiguous (lane*8 + k*512)": (0, table(lambda k, l: l * 8 + k * 512)),
             "b64 contiguous": (2, table(lambda k, l: l * 8 + k * 512)),
             "b128 contiguous (lane*16 + k*1024)": (1, table(lambda k, l: l * 16 + k * 1024)),
             "b128 transposed sub-tile [D][32] (shipped dkdv second product)": (1, table(lambda k, l: (l & 15) * 64 + (((l >> 4) ^ (((l & 15) >> 2) & 3)) << 4) + (k & 7) * 1024))}
    for n, key in KEYS128.items():
        cases[f"D128 tr row-major key={n}"] = (0, table(tr128(key)))
        cases[f"D128 b128 first-product key={n}"] = (1, table(b128_first(key)))
    for n, key in KEYS64.items():
        cases[f"D64 tr row-major key={n}"] = (0, table(tr64(key)))
        cases[f"D64 b128 K-fragment key={n}"] = (1, table(b128_first64(key)))
    for name, (kind, tab) in cases.items():
        res["cycles_per_16_reads"][name] = {f"{w} waves": round(run(kind, tab, w), 1) for w in (1, 4, 8)}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
