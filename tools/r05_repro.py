#!/usr/bin/env python3
"""Do a scene's labels depend on the batch it runs in, or on the run?  N scenes through the engine in different group shapes, several times, against
the single pipeline.  With SG_ENGINE_HASH=1 and a library built with `make DEBUG=1` (the release build carries no debugging code) the engine prints
digests of every phase's device results; the first differing phase of a scene is shown.

    python3 tools/r05_repro.py [N=64] [seed0=40000] [shapes=10x8,6x5,16x1,3x8] [reps=2]          (SG_REPRO_PROFILE=scannet: the ScanNet-shaped scenes)
"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


class Stderr:
    """fd 2 into a file for the duration (the library prints there)"""
    def __enter__(self):
        self.f = tempfile.TemporaryFile(mode="w+b")
        sys.stderr.flush()
        self.saved = os.dup(2)
        os.dup2(self.f.fileno(), 2)
        return self
    def __exit__(self, *a):
        os.dup2(self.saved, 2); os.close(self.saved)
        self.f.seek(0)
        self.text = self.f.read().decode(errors="replace")
        self.f.close()


def digests(text):
    out = {}
    for line in text.splitlines():
        if line.startswith("SGHASH "):
            w = line.split()
            out[(w[1], w[2])] = w[3:]
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
    shapes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[3] if len(sys.argv) > 3 else "10x8,6x5,16x1,3x8").split(",")]
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    pts, segs = (int(v) for v in os.environ.get("SG_REPRO_SIZE", "150000,1500").split(","))
    sem = bool(os.environ.get("SG_REPRO_SEM"))                            # sem_infer (structural threshold 3, weights_g1) instead of ins_infer
    profile = os.environ.get("SG_REPRO_PROFILE", "voronoi")              # "scannet": surfaces, 10k-40k-point floors / walls (the other size classes of FPS, sort, layout)
    jobs = [(pts, segs, seed0 + i, profile, "/tmp/sg_scenes") for i in range(n)]
    it, pool = bench.generate_scenes(jobs, 16)
    from seggroup_amd import hip, weights
    from seggroup_amd.model import Engine, Pipeline
    from seggroup_amd.scene import DeviceScene
    scenes = [DeviceScene.from_synthetic(s, device="cuda:0") for s in it]
    if pool is not None: pool.shutdown()
    if os.environ.get("SG_REPRO_ONLY"):                                    # "i,j,...": only these scenes, repeated to fill n
        keep = [int(v) for v in os.environ["SG_REPRO_ONLY"].split(",")]
        scenes = [scenes[keep[i % len(keep)]] for i in range(n)]
    if os.environ.get("SG_REPRO_LIST_BIG"):
        for i, sc_ in enumerate(scenes):
            big = [(int(k), int(v)) for k, v in enumerate(sc_.h_seg_size) if v > 8192]
            print(f"   scene {i}: segments beyond 8,192 points (index, points): {big}", flush=True)
    W = weights.load_npz(os.path.join(ROOT, "tests", "golden", "weights_g1.npz" if sem else "weights_g2.npz"))
    mode = hip.MODE_SEM_INFER if sem else hip.MODE_INS_INFER
    caps = (max(s.N for s in scenes), max(s.S for s in scenes), max(s.E0 for s in scenes), max(s.V for s in scenes))
    solo = Pipeline(W, *caps, device="cuda:0")
    want = [bench.label_digest(solo.forward(s, mode)) for s in scenes]
    again = [bench.label_digest(solo.forward(s, mode)) for s in scenes]
    print("pipeline twice equal:", want == again, flush=True)
    first = None
    for groups, per in shapes:
        eng = Engine(W, caps, groups=groups, per_group=per, device="cuda:0", timing=0)
        wrong = 0
        events = 0
        tally = {}
        for rep in range(reps):
            with Stderr() as cap:
                got = [bench.label_digest(r) for r in eng.run(scenes, mode)]
            bad = [i for i in range(n) if got[i] != want[i]]
            wrong += len(bad)
            d = digests(cap.text)
            note = ""
            if d:
                if first is None: first = d
                for key in sorted(d, key=lambda k: (k[0], k[1])):
                    if key in first and first[key] != d[key]:
                        fields = [a.split("=")[0] for a, b in zip(d[key], first[key]) if a != b]
                        note += f" [{key[0]} {key[1]}: {','.join(fields)}]"
                        for f_ in fields: tally[(key[1], f_)] = tally.get((key[1], f_), 0) + 1
            rows = [l for l in cap.text.splitlines() if l.startswith("SGROW") or l.startswith("   ")]
            if rows and os.environ.get("SG_SHOW_ROWS"): print("\n".join(rows[:int(os.environ["SG_SHOW_ROWS"])]), flush=True)
            events += note.count("knn")
            if (bad or note) and not os.environ.get("SG_QUIET"): print(f"  engine {groups}x{per} run {rep}: scenes differing from the pipeline {bad}{note}", flush=True)
        print(f"engine {groups}x{per}: {wrong} wrong scene results in {reps} runs of {n}" + (f", {events} kNN tables that differ from the first run's" if first else ""), flush=True)
        if first: print("   digests that differ from the first run's, by (phase, field): " + (", ".join(f"{k[0]}.{k[1]} {v}" for k, v in sorted(tally.items())) or "none"), flush=True)
        eng.close()
        try:
            import ctypes
            fb = (ctypes.c_ulonglong * 8)()
            hip.lib().sg_debug_fps_check(fb)
            print(f"   FPS self-check (chunk-pruned path): {fb[3]} picks checked, {fb[2]} differ from a serial evaluation; half-wave maxima / first indices that a shuffle butterfly computes differently: {fb[0]} / {fb[1]}", flush=True)
        except AttributeError:
            pass
        try:
            import ctypes
            buf = (ctypes.c_ulonglong * 136)()
            hip.lib().sg_debug_knn_check(buf)
            print(f"   kNN self-check: {buf[2]} seeded tiles, {buf[0]} lists out of order after seeding, {buf[1]} at the output stage", flush=True)
            if buf[0]:
                print("   example: row %d point/seed id %d, cluster from %d (%d points), former cluster %d, tile %d" % (buf[48] >> 32, buf[48] & 0xffffffff, buf[49] >> 32, buf[49] & 0xffffffff, buf[50] >> 32, buf[50] & 0xffffffff))
                print("   seed ids : " + " ".join(str(buf[8 + j] >> 32) for j in range(20)))
                print("   positions: " + " ".join(str(ctypes.c_int32(buf[8 + j] & 0xffffffff).value) for j in range(20)))
                print("   keys     : " + " ".join("%x" % buf[28 + j] for j in range(20)))
                import struct
                f = lambda u: struct.unpack("<f", struct.pack("<I", u & 0xffffffff))[0]
                print("   me: %r %r %r w %r (x*x+y*y+z*z = %r); read again: x %r w %r" % (f(buf[52] >> 32), f(buf[52]), f(buf[53] >> 32), f(buf[53]),
                      f(buf[52] >> 32) ** 2 + f(buf[52]) ** 2 + f(buf[53] >> 32) ** 2, f(buf[54] >> 32), f(buf[54])))
                for j in range(20):
                    print("   seed %2d: rec %r %r %r  score again %r" % (j, f(buf[72 + j]), f(buf[92 + j] >> 32), f(buf[92 + j]), f(buf[72 + j] >> 32)))
        except AttributeError:
            pass


if __name__ == "__main__":
    main()
