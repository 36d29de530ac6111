import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/scenes") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
import raytracer_amd as rta
base = open(os.path.join(ROOT, "scenes", "dragon.txt")).read()
text = base + "\nmaterial {\n\tcolor [0 0 0]\n}\n\narea_light {\n\tcolor [20 18 15]\n}\n\ntriangle_mesh {\n\tvertices [ -0.5 -0.5 3.2  0.5 0.5 3.2  0.5 -0.5 3.2  -0.5 -0.5 3.2  -0.5 0.5 3.2  0.5 0.5 3.2 ]\n}\n"
psc = rta.Scene.parse(text).build_bvh()
psc.set_render(1920, 1080, 128)
gpu = rta.Context(0)
gpu.upload(psc)
best = 1e9
for _ in range(3):
    _, _, st = gpu.render_scene(psc, seed=1, samples_per_stream=8, flags=rta.RENDER_TIME_KERNELS, streams=1, want_accum=False)
    best = min(best, st["gpu_ms"])
print("lit dragon 1080p x 128 spp, one stream: %.1f ms; shadow %.1f ms, intersect %.1f, advance %.1f, cells %.1f; shadow rays %d walked %d" % (best, st["shadow_ms"], st["intersect_ms"], st["advance_ms"], st["camera_ms"], st["shadow_rays"], st["walked_rays"]))
