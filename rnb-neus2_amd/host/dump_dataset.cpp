// dump_dataset.cpp — prints what the scene loader (dataset.hpp) hands to rnb_set_dataset as JSON; used by tests/ to check
// the loader against the scene writer and against hand-computed nerf_matrix_to_ngp cases.
#include "dataset.hpp"

#include <cstdint>

int main(int argc, char** argv) {
	if (argc < 2) { std::fprintf(stderr, "usage: dump_dataset <scene dir | transform.json>\n"); return 2; }
	try {
		const hostio::Dataset ds = hostio::load_dataset(argv[1]);
		std::printf("{\"scale\": %.9g, \"offset\": [%.9g, %.9g, %.9g], \"aabb_scale\": %d, \"from_na\": %d, \"from_mitsuba\": %d, \"n2w_s\": %.9g, \"n2w_t\": [%.9g, %.9g, %.9g], \"views\": [",
			ds.scale, ds.offset[0], ds.offset[1], ds.offset[2], ds.aabb_scale, (int)ds.from_na, (int)ds.from_mitsuba, ds.n2w_s, ds.n2w_t[0], ds.n2w_t[1], ds.n2w_t[2]);
		for (size_t i = 0; i < ds.views.size(); ++i) {
			const rnb_view& v = ds.views[i];
			auto fnv = [](const std::vector<uint16_t>& px) { uint64_t h = 1469598103934665603ull; for (uint16_t p : px) { h ^= p; h *= 1099511628211ull; } return h; };
			std::printf("%s{\"width\": %u, \"height\": %u, \"focal_length\": [%.9g, %.9g], \"principal_point\": [%.9g, %.9g], \"xform\": [", i ? ", " : "", v.width, v.height,
				v.focal_length[0], v.focal_length[1], v.principal_point[0], v.principal_point[1]);
			for (int k = 0; k < 12; ++k) std::printf("%s%.9g", k ? ", " : "", v.xform[k]);
			std::printf("], \"normal_fnv\": \"%016llx\", \"albedo_fnv\": \"%016llx\"}", (unsigned long long)fnv(ds.normals[i].rgba), (unsigned long long)fnv(ds.albedos[i].rgba));
		}
		std::printf("]}\n");
	} catch (const std::exception& e) {
		std::fprintf(stderr, "error: %s\n", e.what());
		return 1;
	}
	return 0;
}
