"""cpu_burn.py N SECONDS -- N busy processes for SECONDS (a stand-in for `stress-ng --cpu N`, which the image lacks).
Used to check that bench.py's number survives a loaded host (driver box ran at loadavg 25)."""
import multiprocessing as mp
import sys
import time


def burn(t_end):
    x = 1.0001
    while time.time() < t_end:
        for _ in range(20000):
            x = x * 1.0000001 + 1e-9
    return x


if __name__ == "__main__":
    n, secs = int(sys.argv[1]), float(sys.argv[2])
    t_end = time.time() + secs
    ps = [mp.Process(target=burn, args=(t_end,), daemon=True) for _ in range(n)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
