// Host-side scene handling for the MI355X backend: the scene-file grammar of the
// reference (src/parser.cpp), its BVH construction (src/main.cpp:133-248), the
// camera set-up (src/main.cpp:624-648) and the BMP writer (src/main.cpp:48-81),
// re-designed around structure-of-arrays storage with no fixed capacity.
//
// Everything numeric here must round exactly as the reference does, because the
// node array and the primitive order decide which primitive wins a traversal
// tie: compile without FMA contraction and without fast-math.

#include "host_scene.hpp"

#include <cerrno>
#include <cfloat>
#include <clocale>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace rta {

namespace {

struct F3 {
	float x, y, z;
};

inline F3 sub(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 add(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline F3 mul(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline F3 quot(F3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline F3 neg(F3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float norm(F3 a) { return sqrtf(dot(a, a)); }
inline F3 unit(F3 a) { return quot(a, norm(a)); }
inline F3 cross(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline F3 load3(const float *p) { return {p[0], p[1], p[2]}; }
inline void store3(float *p, F3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// The reference's Min/Max are ternaries (src/common.h:75-81); signed zeros and
// NaNs therefore resolve by argument order, which the node bounds inherit.
inline float lesser(float a, float b) { return a < b ? a : b; }
inline float greater(float a, float b) { return a > b ? a : b; }

constexpr float kPi = 3.14159265f;      // src/common.h:73

// ---------------------------------------------------------------- tokens

enum class Tok { Word, OpenBrace, CloseBrace, OpenBracket, CloseBracket, End };

struct Token {
	Tok kind;
	size_t at, len;
};

// Scanner over the scene text (src/parser.cpp:54-101): blanks separate, '#'
// starts a comment that runs to the end of the line, braces and brackets are
// single-character tokens, anything else up to a blank/brace/bracket/'#' is a
// word.
class Scanner {
public:
	Scanner(const std::string &file, const std::string &text) : file_(file), text_(text) {}

	Token next()
	{
		for (;;) {
			size_t at = pos_;
			char c = peek(pos_);
			if (c == '\0') {
				last_ = at;
				return {Tok::End, at, 0};
			}
			pos_++;
			if (blank(c)) {
				while (blank(peek(pos_)))
					pos_++;
				continue;
			}
			if (c == '#') {
				while (peek(pos_) != '\n' && peek(pos_) != '\0')
					pos_++;
				if (peek(pos_) == '\n')
					pos_++;
				continue;
			}
			last_ = at;
			switch (c) {
			case '{': return {Tok::OpenBrace, at, 1};
			case '}': return {Tok::CloseBrace, at, 1};
			case '[': return {Tok::OpenBracket, at, 1};
			case ']': return {Tok::CloseBracket, at, 1};
			default:
				while (wordchar(peek(pos_)))
					pos_++;
				return {Tok::Word, at, pos_ - at};
			}
		}
	}

	// Error text of the reference: "ERROR:file[offset+1]: msg" (src/parser.cpp:20-30),
	// where offset is the start of the token read last.
	[[noreturn]] void fail(const char *msg) const
	{
		char buf[64];
		snprintf(buf, sizeof buf, "[%zu]: ", last_ + 1);
		throw SceneError{RTA_E_PARSE, "ERROR:" + file_ + buf + msg};
	}

	bool word_is(const Token &t, const char *lit) const
	{
		size_t n = strlen(lit);
		return t.kind == Tok::Word && t.len == n && text_.compare(t.at, n, lit) == 0;
	}

	float number()
	{
		Token t = next();
		if (t.kind != Tok::Word)
			fail("Expected string");
		// strtof must consume the token exactly (src/parser.cpp:151-159)
		const char *s = text_.c_str() + t.at;
		char *end = nullptr;
		errno = 0;
		float v = strtof(s, &end);
		if ((size_t)(end - s) != t.len || errno)
			fail("Invalid number");
		return v;
	}

	// width / height / samples: decimal, 0..65535 (src/parser.cpp:171-183).  The
	// reference then narrows to 16 bits; the value is kept whole here.
	int32_t dimension()
	{
		Token t = next();
		if (t.kind != Tok::Word)
			fail("Expected string");
		const char *s = text_.c_str() + t.at;
		char *end = nullptr;
		errno = 0;
		long v = strtol(s, &end, 10);
		if ((size_t)(end - s) != t.len || errno)
			fail("Invalid integer");
		if (v > 65535 || v < 0)
			fail("Out of bounds integer");
		return (int32_t)v;
	}

	void triple(float out[3])
	{
		if (next().kind != Tok::OpenBracket)
			fail("Expected array");
		out[0] = number();
		out[1] = number();
		out[2] = number();
		if (next().kind != Tok::CloseBracket)
			fail("Expected ']'");
	}

	size_t mark() const { return pos_; }
	void rewind(size_t m) { pos_ = m; }
	void set_last(size_t at) { last_ = at; }
	size_t last() const { return last_; }

private:
	static bool blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }
	static bool wordchar(char c)
	{
		return !(c == '\0' || c == '{' || c == '}' || c == '[' || c == ']' || c == '#' || blank(c));
	}
	char peek(size_t i) const { return i < text_.size() ? text_[i] : '\0'; }

	const std::string &file_;
	const std::string &text_;
	size_t pos_ = 0, last_ = 0;
};

}  // namespace

// ---------------------------------------------------------------- scene

// Defaults of src/parser.cpp:240-258.
HostScene::HostScene()
{
	memset(&settings, 0, sizeof settings);
	settings.up[2] = 1;
	settings.defocus_angle = -1;
	settings.exposure = 1;
	settings.fov = 90;
	settings.width = 512;
	settings.height = 512;
	settings.samples = 20;
	const float grey[6] = {.5f, .5f, .5f, 1, 1.5f, 0};
	materials.assign(grey, grey + 6);
}

void HostScene::add_triangle(const float v[9], int32_t material)
{
	prim_type.push_back(RTA_PRIM_TRIANGLE);
	prim_material.push_back(material);
	prim_light.push_back(-1);
	geom.insert(geom.end(), v, v + 9);
	// flat normal, src/parser.cpp:337-343
	F3 v0 = load3(v), v1 = load3(v + 3), v2 = load3(v + 6);
	F3 n = unit(cross(sub(v1, v0), sub(v2, v0)));
	for (int k = 0; k < 3; k++) {
		normals.push_back(n.x);
		normals.push_back(n.y);
		normals.push_back(n.z);
	}
}

void HostScene::add_sphere(const float center[3], float radius, int32_t material)
{
	prim_type.push_back(RTA_PRIM_SPHERE);
	prim_material.push_back(material);
	prim_light.push_back(-1);
	const float g[9] = {center[0], center[1], center[2], radius, 0, 0, 0, 0, 0};
	geom.insert(geom.end(), g, g + 9);
	normals.insert(normals.end(), 9, 0.f);
}

HostScene *parse_scene(const std::string &name, const std::string &text)
{
	setlocale(LC_NUMERIC, "C");        // strtof must read '.', src/main.cpp:596
	HostScene *sc = new HostScene();
	try {
		Scanner in(name, text);
		int32_t material = 0;            // sticky: applies to every later primitive
		bool emit = false;               // sticky: once set every later triangle is a light
		float emit_color[3] = {0, 0, 0};

		for (;;) {
			Token cmd = in.next();
			if (cmd.kind == Tok::End)
				break;
			if (cmd.kind != Tok::Word)
				in.fail("Expected command");
			if (in.next().kind != Tok::OpenBrace)
				in.fail("Expected '{'");
			in.set_last(cmd.at);          // command-level errors point at the command

			auto each_key = [&](auto &&on_key) {
				for (;;) {
					Token key = in.next();
					if (key.kind == Tok::CloseBrace)
						return;
					if (key.kind != Tok::Word)
						in.fail("Expected key");
					if (!on_key(key))
						in.fail("Unknown key");
				}
			};

			if (in.word_is(cmd, "sphere")) {
				float center[3] = {0, 0, 0}, radius = 0;
				each_key([&](const Token &k) {
					if (in.word_is(k, "radius")) radius = in.number();
					else if (in.word_is(k, "center")) in.triple(center);
					else return false;
					return true;
				});
				sc->add_sphere(center, radius, material);
			} else if (in.word_is(cmd, "triangle_mesh")) {
				each_key([&](const Token &k) {
					if (!in.word_is(k, "vertices"))
						return false;
					if (in.next().kind != Tok::OpenBracket)
						in.fail("Expected array");
					for (;;) {
						size_t m = in.mark();
						if (in.next().kind == Tok::CloseBracket)
							break;
						in.rewind(m);
						float v[9];
						for (float &f : v)
							f = in.number();
						sc->add_triangle(v, material);
						if (emit) {
							sc->prim_light.back() = (int32_t)sc->light_prim.size();
							sc->light_color.insert(sc->light_color.end(), emit_color, emit_color + 3);
							sc->light_prim.push_back(-1);     // linked after the BVH build
						}
					}
					return true;
				});
			} else if (in.word_is(cmd, "material")) {
				float m[6] = {.7f, .7f, .7f, 1, 1.5f, 0};
				each_key([&](const Token &k) {
					if (in.word_is(k, "color")) in.triple(m);
					else if (in.word_is(k, "roughness")) {
						float r = in.number();
						m[3] = r * r;             // alpha = r^2, src/parser.cpp:368-369
					} else if (in.word_is(k, "ior")) m[4] = in.number();
					else if (in.word_is(k, "metallic")) m[5] = in.number();
					else return false;
					return true;
				});
				material = (int32_t)(sc->materials.size() / 6);
				sc->materials.insert(sc->materials.end(), m, m + 6);
			} else if (in.word_is(cmd, "area_light")) {
				emit = true;
				each_key([&](const Token &k) {
					if (!in.word_is(k, "color"))
						return false;
					in.triple(emit_color);
					return true;
				});
			} else if (in.word_is(cmd, "render")) {
				rta_scene_settings &s = sc->settings;
				each_key([&](const Token &k) {
					if (in.word_is(k, "camera")) in.triple(s.camera);
					else if (in.word_is(k, "look_at")) in.triple(s.look_at);
					else if (in.word_is(k, "up")) in.triple(s.up);
					else if (in.word_is(k, "fov")) s.fov = in.number();
					else if (in.word_is(k, "defocus_angle")) s.defocus_angle = in.number();
					else if (in.word_is(k, "exposure")) s.exposure = in.number();
					else if (in.word_is(k, "width")) s.width = in.dimension();
					else if (in.word_is(k, "height")) s.height = in.dimension();
					else if (in.word_is(k, "samples")) s.samples = in.dimension();
					else if (in.word_is(k, "sky_box_color")) in.triple(s.sky);
					else return false;
					return true;
				});
			} else {
				in.fail("Unknown command");
			}
		}
	} catch (...) {
		delete sc;
		throw;
	}
	return sc;
}

HostScene *load_scene(const std::string &path)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f)
		throw SceneError{RTA_E_IO, "Couldn't open '" + path + "'"};
	std::string text;
	char buf[1 << 16];
	size_t got;
	while ((got = fread(buf, 1, sizeof buf, f)) > 0)
		text.append(buf, got);
	fclose(f);
	// the reference treats the buffer as NUL-terminated (src/parser.cpp:271)
	size_t nul = text.find('\0');
	if (nul != std::string::npos)
		text.resize(nul);
	return parse_scene(path, text);
}

// ---------------------------------------------------------------- BVH

namespace {

struct Builder {
	HostScene &sc;
	std::vector<float> plo, phi, centroid;   // per primitive (file order), 3 floats each
	std::vector<int32_t> &order;             // BVH position -> file position

	explicit Builder(HostScene &s) : sc(s), order(s.perm)
	{
		int32_t n = sc.prim_count();
		plo.resize((size_t)n * 3);
		phi.resize((size_t)n * 3);
		centroid.resize((size_t)n * 3);
		for (int32_t k = 0; k < n; k++) {
			const float *g = &sc.geom[(size_t)k * 9];
			F3 lo, hi, c;
			if (sc.prim_type[k] == RTA_PRIM_SPHERE) {
				F3 ctr = load3(g);
				float r = g[3];
				lo = {ctr.x - r, ctr.y - r, ctr.z - r};
				hi = {ctr.x + r, ctr.y + r, ctr.z + r};
				c = ctr;
			} else {
				F3 v0 = load3(g), v1 = load3(g + 3), v2 = load3(g + 6);
				// src/main.cpp:156-162: start from v0, fold v1 then v2
				lo = v0;
				lo = {lesser(lo.x, v1.x), lesser(lo.y, v1.y), lesser(lo.z, v1.z)};
				lo = {lesser(lo.x, v2.x), lesser(lo.y, v2.y), lesser(lo.z, v2.z)};
				hi = v0;
				hi = {greater(hi.x, v1.x), greater(hi.y, v1.y), greater(hi.z, v1.z)};
				hi = {greater(hi.x, v2.x), greater(hi.y, v2.y), greater(hi.z, v2.z)};
				// src/main.cpp:212
				c = mul(add(add(v0, v1), v2), 1.f / 3.f);
			}
			store3(&plo[(size_t)k * 3], lo);
			store3(&phi[(size_t)k * 3], hi);
			store3(&centroid[(size_t)k * 3], c);
		}
	}

	int32_t new_node(int32_t first, int32_t count)
	{
		int32_t id = sc.node_total();
		sc.node_index.push_back(first);
		sc.node_count.push_back(count);
		sc.node_bounds.insert(sc.node_bounds.end(), 6, 0.f);
		fit(id);
		return id;
	}

	// src/main.cpp:133-168: fold the primitives' boxes in BVH order
	void fit(int32_t id)
	{
		float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
		int32_t first = sc.node_index[id], last = first + sc.node_count[id];
		for (int32_t i = first; i < last; i++) {
			const float *a = &plo[(size_t)order[i] * 3], *b = &phi[(size_t)order[i] * 3];
			for (int ax = 0; ax < 3; ax++) {
				lo[ax] = lesser(lo[ax], a[ax]);
				hi[ax] = greater(hi[ax], b[ax]);
			}
		}
		float *dst = &sc.node_bounds[(size_t)id * 6];
		memcpy(dst, lo, 12);
		memcpy(dst + 3, hi, 12);
	}

	// src/main.cpp:170-248
	void run()
	{
		int32_t n = sc.prim_count();
		order.resize(n);
		std::iota(order.begin(), order.end(), 0);
		sc.node_bounds.clear();
		sc.node_index.clear();
		sc.node_count.clear();
		sc.node_bounds.reserve((size_t)n * 12);
		sc.node_index.reserve((size_t)n * 2);
		sc.node_count.reserve((size_t)n * 2);

		int32_t cur = new_node(0, n);
		// Pending second children.  The reference keeps 64 and, when they are all
		// in use, gives up on the node at hand and resumes from the newest pending
		// one (src/main.cpp:182-185); the same rule is kept so that deep inputs
		// produce the same tree.
		int32_t pending[64];
		int depth = 0;

		for (;;) {
			if (depth >= 64) {
				cur = pending[--depth];
				continue;
			}
			if (sc.node_count[cur] > 2) {
				const float *b = &sc.node_bounds[(size_t)cur * 6];
				float ext[3] = {b[3] - b[0], b[4] - b[1], b[5] - b[2]};
				int axis = 0;
				if (ext[1] > ext[0])
					axis = 1;
				if (ext[2] > ext[axis])
					axis = 2;
				float split = b[axis] + ext[axis] * .5f;

				// two-pointer partition: centroids below the split plane stay in
				// front, the others are swapped to the shrinking tail
				int32_t first = sc.node_index[cur], count = sc.node_count[cur];
				int32_t i = first, j = first + count - 1;
				while (i <= j) {
					if (centroid[(size_t)order[i] * 3 + axis] < split) {
						i++;
					} else {
						std::swap(order[i], order[j]);
						j--;
					}
				}
				int32_t left = i - first;
				if (left != 0 && left != count) {
					int32_t l = new_node(first, left);
					int32_t r = new_node(i, count - left);
					sc.node_index[cur] = l;
					sc.node_count[cur] = 0;
					pending[depth++] = r;
					cur = l;
					continue;
				}
			}
			// cur stays a leaf (small enough, or all centroids on one side)
			if (depth == 0)
				break;
			cur = pending[--depth];
		}
	}
};

template <typename T>
void permute_rows(std::vector<T> &a, const std::vector<int32_t> &order, size_t width)
{
	std::vector<T> out(a.size());
	for (size_t i = 0; i < order.size(); i++)
		memcpy(&out[i * width], &a[(size_t)order[i] * width], width * sizeof(T));
	a.swap(out);
}

// Rows into BVH order; lights point at their primitive's new place (src/main.cpp:617-622).
void finish_build(HostScene &sc)
{
	permute_rows(sc.prim_type, sc.perm, 1);
	permute_rows(sc.prim_material, sc.perm, 1);
	permute_rows(sc.prim_light, sc.perm, 1);
	permute_rows(sc.geom, sc.perm, 9);
	permute_rows(sc.normals, sc.perm, 9);
	for (int32_t i = 0; i < sc.prim_count(); i++)
		if (sc.prim_light[i] >= 0)
			sc.light_prim[sc.prim_light[i]] = i;
}

}  // namespace

void HostScene::build_bvh()
{
	if (prim_count() == 0)
		throw SceneError{RTA_E_INVALID, "scene has no primitives"};
	if (!perm.empty()) {
		// already built once: the arrays are in BVH order; rebuilding from that
		// order would give a different (though valid) tree than the file order
		throw SceneError{RTA_E_INVALID, "BVH already built for this scene"};
	}
	Builder b(*this);
	b.run();
	finish_build(*this);
}

void HostScene::adopt_tree(std::vector<int32_t> &&order, std::vector<float> &&bounds, std::vector<int32_t> &&index,
                           std::vector<int32_t> &&count)
{
	if (!perm.empty())
		throw SceneError{RTA_E_INVALID, "BVH already built for this scene"};
	if ((int32_t)order.size() != prim_count() || index.size() != count.size() || bounds.size() != index.size() * 6)
		throw SceneError{RTA_E_INVALID, "tree does not fit the scene"};
	perm = std::move(order);
	node_bounds = std::move(bounds);
	node_index = std::move(index);
	node_count = std::move(count);
	finish_build(*this);
}

void HostScene::describe(rta_scene_desc *out) const
{
	out->n_prims = prim_count();
	out->n_nodes = node_total();
	out->n_materials = (int32_t)(materials.size() / 6);
	out->n_lights = (int32_t)light_prim.size();
	out->prim_type = prim_type.data();
	out->prim_material = prim_material.data();
	out->prim_light = prim_light.data();
	out->prim_geom = geom.data();
	out->prim_normals = normals.data();
	out->node_bounds = node_bounds.data();
	out->node_index = node_index.data();
	out->node_count = node_count.data();
	out->materials = materials.data();
	out->light_color = light_color.data();
	out->light_prim = light_prim.data();
}

// ---------------------------------------------------------------- camera, BMP

// src/main.cpp:624-648
void camera_from_settings(const rta_scene_settings &s, rta_camera *out)
{
	F3 eye = load3(s.camera), target = load3(s.look_at), world_up = load3(s.up);
	float focus = norm(sub(target, eye));
	float fov_r = s.fov * kPi / 180;
	float aspect = (float)s.width / (float)s.height;
	float vh = 2 * tanf(fov_r / 2) * focus;
	float vw = vh * aspect;

	F3 fwd = unit(sub(target, eye));
	F3 right = unit(cross(fwd, world_up));
	F3 up = cross(right, fwd);
	F3 span_u = mul(right, vw);
	F3 span_v = mul(neg(up), vh);

	store3(out->origin, eye);
	store3(out->upper_left, sub(sub(add(eye, mul(fwd, focus)), quot(span_u, 2)), quot(span_v, 2)));
	store3(out->du, quot(span_u, (float)s.width));
	store3(out->dv, quot(span_v, (float)s.height));

	float defocus_r = s.defocus_angle * kPi / 180;
	float lens = focus * tanf(defocus_r / 2);
	store3(out->lens_u, mul(right, lens));
	store3(out->lens_v, mul(up, lens));
	out->defocus_angle = s.defocus_angle;
}

// src/main.cpp:48-81
int write_bmp(const char *path, int32_t w, int32_t h, const uint8_t *bgr)
{
	FILE *f = fopen(path, "wb");
	if (!f)
		return RTA_E_IO;
	const uint32_t row_bytes = (uint32_t)w * 3;
	const uint32_t pad = (4 - row_bytes % 4) % 4;
	const uint32_t file_size = 54 + 3 * (uint32_t)w * (uint32_t)h;    // the reference leaves padding out of this field
	uint8_t head[54] = {0};
	auto put32 = [&](int at, uint32_t v) {
		for (int k = 0; k < 4; k++)
			head[at + k] = (uint8_t)(v >> (8 * k));
	};
	head[0] = 'B';
	head[1] = 'M';
	put32(2, file_size);
	put32(10, 54);
	put32(14, 40);
	put32(18, (uint32_t)w);
	put32(22, (uint32_t)h);
	head[26] = 1;
	head[28] = 24;
	bool ok = fwrite(head, 1, 54, f) == 54;
	const uint8_t zeros[3] = {0, 0, 0};
	for (int32_t row = h - 1; row >= 0 && ok; row--) {
		ok = fwrite(bgr + (size_t)row * row_bytes, 1, row_bytes, f) == row_bytes;
		if (pad)
			ok = ok && fwrite(zeros, 1, pad, f) == pad;
	}
	ok = (fclose(f) == 0) && ok;
	return ok ? RTA_OK : RTA_E_IO;
}

}  // namespace rta
