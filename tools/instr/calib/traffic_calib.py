"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS path's access shapes (verdict r05, item 5).

The guide calibrates FETCH_SIZE only for wide coalesced reads (x2: 128-byte requests tallied at 64).  The blend kernels
gather 48-byte records, read 1-KB rows of 16 bytes per lane and scatter 48-byte records + 4-byte tags.  This driver launches
kernels that move a KNOWN number of bytes in exactly those shapes (tools/instr/calib/traffic_calib.hip) over tables of 2-3 GB
(far beyond L2 + the 256 MB Infinity Cache, every element touched once per launch); run it under rocprofv3 --pmc and compare:

    tools/instr/calib/run_calib.sh            (on the GPU box; writes profiles/r06_counter_calibration.txt)

Standalone: uses its own little library, not libmanus_hip.so."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libtraffic_calib.so")


def build():
    src = os.path.join(HERE, "traffic_calib.hip")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", SO, src])
    return SO


# pattern -> (kernel name in the trace, useful bytes read per launch, useful bytes written per launch)
N_REC = 1 << 25            # 33.5 M records of 48 bytes = 1.6 GB table, every record once
N_ROWS = 1 << 21           # 2.1 M rows of 1 KB = 2.1 GB
STREAM = 3 << 30
PATTERNS = {
    "cal_stream_read": (STREAM, 0),
    "cal_stream_write": (0, STREAM),
    "cal_gather48": (48 * N_REC, 0),
    "cal_gather48_indexed": (52 * N_REC, 0),
    "cal_rows16": (1024 * N_ROWS, 0),
    "cal_scatter48": (0, 48 * N_REC),            # launched twice per repetition: without / with the tag store (+4 B); see main
    "cal_scatter4": (0, 4 * N_REC),
}


def main(reps=3):
    L = ctypes.CDLL(build())
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    big = torch.zeros(STREAM // 4, dtype=torch.float32, device=dev)
    sink = torch.zeros(16, dtype=torch.float32, device=dev)
    tags = torch.zeros(N_REC, dtype=torch.int32, device=dev)
    idx = ((torch.arange(N_REC, dtype=torch.int64, device=dev) * 2654435761) & (N_REC - 1)).to(torch.int32)
    u32 = ctypes.c_uint32
    for _ in range(reps):
        L.run_stream_read(p(big), p(sink), ctypes.c_size_t(STREAM), st)
        L.run_stream_write(p(big), ctypes.c_size_t(STREAM), st)
        L.run_gather48(p(big), p(sink), u32(N_REC), u32(N_REC - 1), st)
        L.run_gather48_indexed(p(big), p(idx), p(sink), u32(N_REC), st)
        L.run_rows16(p(big), p(sink), u32(N_ROWS), u32(N_ROWS - 1), st)
        L.run_scatter48(p(big), p(tags), u32(N_REC), u32(N_REC - 1), 0, st)
        L.run_scatter48(p(big), p(tags), u32(N_REC), u32(N_REC - 1), 1, st)
        L.run_scatter4(p(tags), u32(N_REC), u32(N_REC - 1), st)
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        print(build())
    else:
        main()
