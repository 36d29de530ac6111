// `raytracer FILE` -> image.bmp, the reference's command-line contract
// (src/main.cpp:594-605,717), rendered on MI355X GPUs through the C ABI.
//
//   raytracer scene.txt                       one GPU, writes ./image.bmp
//   raytracer scene.txt -o out.bmp --gpus 4   tiles interleaved over 4 GPUs
//   --width W --height H --samples N          override the file's render block
//   --seed S --samples-per-stream K           random-stream layout
//
// Exit status 1 with "Usage: ..." when no file is given, and 1 with the
// reference's "ERROR:file[pos]: msg" text on a scene error.

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "raytracer_amd.h"

namespace {

struct Options {
	const char *scene = nullptr;
	std::string out = "image.bmp";
	int gpus = 1;
	int width = -1, height = -1, samples = -1;
	unsigned seed = 69420;            // the reference's seed base (src/main.cpp:659)
	int samples_per_stream = 32;
};

bool parse_args(int argc, char **argv, Options &o)
{
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		auto value = [&](int &dst) {
			if (i + 1 >= argc)
				return false;
			dst = atoi(argv[++i]);
			return true;
		};
		if (a == "-o") {
			if (i + 1 >= argc)
				return false;
			o.out = argv[++i];
		} else if (a == "--gpus") {
			if (!value(o.gpus)) return false;
		} else if (a == "--width") {
			if (!value(o.width)) return false;
		} else if (a == "--height") {
			if (!value(o.height)) return false;
		} else if (a == "--samples") {
			if (!value(o.samples)) return false;
		} else if (a == "--samples-per-stream") {
			if (!value(o.samples_per_stream)) return false;
		} else if (a == "--seed") {
			int s;
			if (!value(s)) return false;
			o.seed = (unsigned)s;
		} else if (!a.empty() && a[0] == '-') {
			return false;
		} else if (!o.scene) {
			o.scene = argv[i];
		} else {
			return false;
		}
	}
	return o.scene != nullptr && o.gpus >= 1;
}

}  // namespace

int main(int argc, char **argv)
{
	Options opt;
	if (!parse_args(argc, argv, opt)) {
		fprintf(stderr, "Usage: %s [FILE]\n", argv[0]);
		return 1;
	}

	char err[1024] = "";
	rta_scene *scene = nullptr;
	int rc = rta_scene_load(opt.scene, &scene, err, sizeof err);
	if (rc != RTA_OK) {
		fprintf(stderr, "%s\n", err);
		return 1;
	}
	// the reference builder's tree, built on the first GPU (on the host if that fails)
	rta_context *builder = nullptr;
	rc = rta_create(0, &builder) == RTA_OK ? rta_scene_build_bvh_device(builder, scene, nullptr) : RTA_E_NO_DEVICE;
	if (builder)
		rta_destroy(builder);
	if (rc != RTA_OK && (rc = rta_scene_build_bvh(scene)) != RTA_OK) {
		fprintf(stderr, "ERROR: %s\n", rta_global_error());
		return 1;
	}
	rta_scene_settings set;
	rta_scene_get_settings(scene, &set);
	if (opt.width > 0) set.width = opt.width;
	if (opt.height > 0) set.height = opt.height;
	if (opt.samples > 0) set.samples = opt.samples;
	if (set.width <= 0 || set.height <= 0 || set.samples <= 0) {
		fprintf(stderr, "ERROR: width, height and samples must be positive\n");
		return 1;
	}
	rta_scene_desc desc;
	rta_camera cam;
	rta_scene_describe(scene, &desc);
	rta_camera_from_settings(&set, &cam);

	const int have = rta_device_count();
	if (have < opt.gpus) {
		fprintf(stderr, "ERROR: %d GPU(s) requested, %d present\n", opt.gpus, have);
		return 1;
	}

	const size_t px = (size_t)set.width * set.height;
	std::vector<uint8_t> image(px * 3, 0);
	std::vector<std::string> failures(opt.gpus);
	std::vector<rta_stats> stats(opt.gpus);
	fprintf(stderr, "Raytracing on %d GPU(s)...\n", opt.gpus);

	auto work = [&](int g) {
		rta_context *ctx = nullptr;
		if (rta_create(g, &ctx) != RTA_OK) {
			failures[g] = rta_global_error();
			return;
		}
		rta_render_params p;
		memset(&p, 0, sizeof p);
		p.width = set.width;
		p.height = set.height;
		p.samples = set.samples;
		p.exposure = set.exposure;
		memcpy(p.sky, set.sky, sizeof p.sky);
		p.seed = opt.seed;
		p.samples_per_stream = opt.samples_per_stream;
		p.shard_index = g;
		p.shard_count = opt.gpus;
		p.out_mem = RTA_MEM_HOST;
		// every shard writes disjoint pixels of the shared host image
		if (rta_upload_scene(ctx, &desc) != RTA_OK ||
		    rta_render(ctx, &cam, &p, nullptr, image.data(), &stats[g]) != RTA_OK)
			failures[g] = rta_last_error(ctx);
		rta_destroy(ctx);
	};
	std::vector<std::thread> threads;
	for (int g = 1; g < opt.gpus; g++)
		threads.emplace_back(work, g);
	work(0);
	for (auto &t : threads)
		t.join();
	for (int g = 0; g < opt.gpus; g++)
		if (!failures[g].empty()) {
			fprintf(stderr, "ERROR: GPU %d: %s\n", g, failures[g].c_str());
			return 1;
		}

	double ms = 0;
	unsigned long long rays = 0, samples = 0;
	for (const rta_stats &s : stats) {
		ms = s.gpu_ms > ms ? s.gpu_ms : ms;
		rays += s.rays + s.shadow_rays;
		samples += s.samples;
	}
	fprintf(stderr, "%llu samples, %llu rays in %.1f ms: %.1f Msamples/s, %.1f Mrays/s\n", samples, rays, ms,
	        samples / ms / 1e3, rays / ms / 1e3);

	if (rta_write_bmp(opt.out.c_str(), set.width, set.height, image.data()) != RTA_OK) {
		fprintf(stderr, "ERROR: %s\n", rta_global_error());
		return 1;
	}
	rta_scene_free(scene);
	return 0;
}
