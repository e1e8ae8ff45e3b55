#!/usr/bin/env python3
"""Which CUs / XCDs does a CU-masked stream use?  Launches xvit_cu_trace (lingering blocks) on streams with
different masks and prints the XCDs and the number of distinct CUs observed."""
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import _lib, cu_mask  # noqa: E402


def trace(stream, nblocks=2048, linger_us=30):
    out = torch.zeros(nblocks * 2, dtype=torch.int32, device="cuda:0")
    with torch.cuda.stream(stream):
        _lib.check(_lib.load().xvit_cu_trace(out.data_ptr(), nblocks, linger_us, stream.cuda_stream), "xvit_cu_trace")
    stream.synchronize()
    o = out.cpu().view(nblocks, 2)
    xcc = o[:, 0] & 0xF
    hw = o[:, 1]
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    places = Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcd = Counter(x for (x, _, _, _) in places)
    first = xcc[:16].tolist()
    return len(places), dict(sorted(per_xcd.items())), first


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    print("CUs:", n_cu)
    print("default stream          :", trace(torch.cuda.Stream(dev)))
    for name, bits in (("bits 0..127", range(128)), ("bits 128..255", range(128, 256)), ("even bits", range(0, 256, 2)),
                       ("bits with b%8<4", [b for b in range(256) if b % 8 < 4]), ("bits 0..31", range(32)), ("bits b%8==0", range(0, 256, 8))):
        st = cu_mask.masked_stream(dev, bits)
        print(f"{name:24s}:", trace(st))


if __name__ == "__main__":
    main()
