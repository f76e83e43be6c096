"""TEST TOOLING (build container only): the compiled reference behind a pipe.

Reads `<variant> <recipient_size> <hex of a raw DEFLATE stream>` lines on stdin, answers
`<good> <final or -> <sha256 of the output bytes>` per line.  It lives in its own process so that an
input on which the reference aborts, loops or faults costs one skipped case, not the test run
(tests/test_oracle_vs_reference.py, tests/golden/make_golden.py)."""
import hashlib
import os
import random
import signal
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding  # noqa: E402


class RefWorker:
    """the reference in a child process (tests/ref_worker.py): a crash is a skipped case"""

    def __init__(self):
        self.p = None
        self.crashes = 0

    def _start(self):
        import subprocess
        import sys

        self.p = subprocess.Popen([sys.executable, os.path.abspath(__file__)],
                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)

    def inflate(self, variant, raw, cap):
        """-> (good, final, sha256) or None when the reference died on this input"""
        if self.p is None or self.p.poll() is not None:
            self._start()
        try:
            self.p.stdin.write(f"{variant} {cap} {raw.hex()}\n")
            self.p.stdin.flush()
            ans = self.p.stdout.readline().split()
        except (BrokenPipeError, OSError):
            ans = []
        if len(ans) != 3:
            self.crashes += 1
            self.p.kill()
            self.p.wait()
            self.p = None
            return None
        return int(ans[0]), (None if ans[1] == "-" else int(ans[1])), ans[2]

    def close(self):
        if self.p is not None and self.p.poll() is None:
            self.p.stdin.close()
            self.p.wait(timeout=30)


def corrupt_cases(seed, count, max_plain=2500):
    """damaged raw DEFLATE streams, the same mutators as the GPU suite (truncation, bit flips,
    spliced bytes, appended bytes) over zlib streams of every strategy -> [(raw, recipient_size)].
    recipient_size is far beyond anything the stream can produce: asserts-off SILENCE builds of the
    reference have no output bound at all (Q12)."""
    rng = random.Random(seed)
    bases = []
    for k in range(60):
        n = rng.randint(50, max_plain)
        kind = rng.randrange(4)
        if kind == 0:
            data = bytes(rng.getrandbits(8) for _ in range(n))
        elif kind == 1:
            words = [bytes(rng.getrandbits(8) for _ in range(rng.randint(3, 9))) for _ in range(40)]
            data = (b" ".join(rng.choice(words) for _ in range(n // 5 + 1)))[:n]
        elif kind == 2:
            data = bytes(rng.choice(b"ab") for _ in range(n))
        else:
            data = bytes(rng.choice(b"abcdefgh \n") for _ in range(n))
        c = zlib.compressobj(rng.choice([1, 6, 9]), zlib.DEFLATED, -15, 9,
                             rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE]))
        raw = c.compress(data)
        if rng.random() < 0.3:
            raw += c.flush(zlib.Z_FULL_FLUSH)
        bases.append(raw + c.flush())
    out = []
    for it in range(count):
        raw = bytearray(bases[it % len(bases)])
        mode = rng.randrange(4)
        if mode == 0:
            raw = raw[: rng.randint(5, len(raw))]
        elif mode == 1:
            for _ in range(rng.randint(1, 4)):
                raw[rng.randrange(len(raw))] ^= 1 << rng.randrange(8)
        elif mode == 2:
            i = rng.randrange(len(raw))
            raw[i:i + rng.randint(1, 8)] = bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 8)))
        else:
            raw = raw + bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 64)))
        out.append((bytes(raw), 1100 * len(raw) + 4096))
    return out


def main():
    refs = {}
    for line in sys.stdin:
        variant, cap, hx = line.split()
        if variant not in refs:
            refs[variant] = binding.Reference(variant)
        signal.alarm(20)  # a looping reference kills the worker: the parent sees EOF
        good, final, out = refs[variant].inflate(bytes.fromhex(hx), int(cap))
        signal.alarm(0)
        sys.stdout.write(f"{int(good)} {'-' if final is None else final} {hashlib.sha256(out).hexdigest()}\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
