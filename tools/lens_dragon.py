"""dragon.txt through a lens at 1080p: frame time, and what share of the rays the walk took."""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "tests")]
import orc
import raytracer_amd as rta

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
text = orc.scene_variant("dragon.txt", 1920, 1080, spp).replace("look_at [0 0 1]", "look_at [0 0 1]\n\tdefocus_angle 2.0")
psc = rta.Scene.parse(text).build_bvh()
gpu = rta.Context(0)
gpu.upload(psc)
for k in range(3):
    t0 = time.perf_counter()
    bgr, acc, st = gpu.render_scene(psc, seed=5, samples_per_stream=8)
    dt = time.perf_counter() - t0
    print(f"{dt * 1e3:.1f} ms  rays {st['rays']}  early {st['early_rays']}  walked {st['walked_rays']}  retraced {st.get('retraced_rays')}", flush=True)
