// testbed_main.cpp — the `./build/testbed` command line of the reference (src/main.cu:73-472), host side in C++ over the
// C-ABI of include/rnb_neus2.h. Same flags, defaults, exit codes, output paths and stdout progress line; the training hot
// path runs in librnb_neus2_hip.so. No GUI (the pipeline always passes --no-gui, rnb_neus2/pipeline.py:37).
//
//   testbed --scene <dir>/ --maxiter N --no-gui --mask-weight F [--save-snapshot] [--save-mesh --resolution R]
//           [--snapshot PATH] [--opti-lights] [--no-albedo] [--free-memory] [--lone] [--supernormal] [--no-rgbplus]
//           [--relu] [--bce] [--disable-snap-to-center] [-n/-c/--network/--config CFG] [--no-train] [--save-each N]
//           [--fractional-training N] [--width W] [--height H] [-v/--version] [-h/--help]
#include "../../include/rnb_neus2.h"
#include "dataset.hpp"
#include "json_min.hpp"
#include "mesh.hpp"
#include "msgpack_min.hpp"
#include "png16.hpp"

#include "dist_transport.hpp"

#ifdef RNB_WITH_HIP // the build over librnb_neus2_hip.so (the CPU-checker build of this file, tests/, has no device)
#include <hip/hip_runtime_api.h>
#endif
#ifdef RNB_WITH_RCCL // several processes, one per GPU, exchange counters and gradients over RCCL (no reference counterpart)
#include <rccl/rccl.h>
#endif

#include <sys/stat.h>
#include <dirent.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#define NGP_VERSION "rnb-neus2-mi355x 0.1"

namespace {

using namespace hostio;
std::string exe_dir() {
	char buf[4096];
	ssize_t n = ::readlink("/proc/self/exe", buf, sizeof(buf) - 1);
	if (n <= 0) return ".";
	buf[n] = 0;
	return parent_path(buf);
}

// ---- args (dependencies/args semantics as used by src/main.cu:83-258) ----
struct FlagSpec { std::vector<std::string> names; bool has_value; const char* meta; const char* help; };
const std::vector<FlagSpec> FLAGS = {
	{{"h", "help"}, false, "HELP", "Display this help menu."},
	{{"lone"}, false, "L_ONE", "Activate l_one between colors !"},
	{{"supernormal"}, false, "SUPERNORMAL", "Activate Supernormal loss function"},
	{{"no-rgbplus"}, false, "NO_RGB_PLUS", "Deactivate rgb normalisation"},
	{{"disable-snap-to-center"}, false, "DISABLE_SNAP_TO_CENTER", "Disable snap to center for the camera !"},
	{{"relu"}, false, "RELU", "Activate ReLU for shading !"},
	{{"bce"}, false, "BCE", "Apply BCE mask loss instead of Sigmoid BCE !"},
	{{"n", "c", "network", "config"}, true, "CONFIG", "Path to the network config. Uses the scene's default if unspecified."},
	{{"no-gui"}, false, "NO_GUI", "Disables the GUI and instead reports training progress on the command line."},
	{{"save-mesh"}, false, "SAVE_MESH", "Save as a mesh when it's done."},
	{{"save-snapshot"}, false, "SAVE_SNAPSHOT", "Save as a snapshot when it's done."},
	{{"opti-lights"}, false, "OPTI-LIGHTS", "Use optimal lights per pixels"},
	{{"no-albedo"}, false, "no-albedo", "To use when you don't want to optimize the albedo"},
	{{"no-train"}, false, "NO_TRAIN", "Disables training on startup."},
	{{"free-memory"}, false, "FREE-MEMORY", "Free images from GPU memory"},
	{{"s", "scene"}, true, "SCENE", "The scene to load (directory with transform*.json)."},
	{{"snapshot"}, true, "SNAPSHOT", "Optional snapshot to load upon startup."},
	{{"width"}, true, "WIDTH", "Resolution width of the GUI."},
	{{"save-each"}, true, "SAVE_EACH", "Save mesh each X number of iterations"},
	{{"resolution"}, true, "RESOLUTION", "Resolution used for marching cube"},
	{{"height"}, true, "HEIGHT", "Resolution height of the GUI."},
	{{"maxiter"}, true, "MAXITER", "Maximum number of iterations."},
	{{"v", "version"}, false, "VERSION", "Display the version of neural graphics primitives."},
	{{"mask-weight"}, true, "MASK_WEIGHT", "Mask weight."},
	{{"fractional-training"}, true, "FRACTIONAL_TRAINING", "Step for fractional training"},
	{{"accumulate"}, true, "ACCUMULATE", "fp32 (default) or half: width of the accumulators (half = the reference's arithmetic as coded). Not a flag of the reference."},
	{{"deterministic"}, false, "", "Sum the hash-grid gradients as fixed-point integers: a bit-reproducible training run. Not a flag of the reference."},
};

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };
struct ValidationError : std::runtime_error { using std::runtime_error::runtime_error; };

struct Args {
	std::map<std::string, std::string> values; // canonical (last) name -> value ("" for plain flags)
	bool has(const std::string& k) const { return values.count(k) > 0; }
	const std::string& get(const std::string& k) const { return values.at(k); }
	uint32_t get_u32(const std::string& k) const {
		const std::string& s = get(k);
		char* e = nullptr;
		unsigned long long v = std::strtoull(s.c_str(), &e, 10);
		if (s.empty() || *e || s[0] == '-') throw ParseError("Argument '" + k + "' received invalid value type '" + s + "'");
		return (uint32_t)v;
	}
	float get_f32(const std::string& k) const {
		const std::string& s = get(k);
		char* e = nullptr;
		float v = std::strtof(s.c_str(), &e);
		if (s.empty() || *e) throw ParseError("Argument '" + k + "' received invalid value type '" + s + "'");
		return v;
	}
};

void print_help(std::ostream& os, const char* prog) {
	os << "  " << prog << " {OPTIONS}\n\n    neural graphics primitives\n    version " NGP_VERSION "\n\n  OPTIONS:\n\n";
	for (const auto& f : FLAGS) {
		os << "      ";
		for (size_t i = 0; i < f.names.size(); ++i) { os << (i ? ", " : "") << (f.names[i].size() == 1 ? "-" : "--") << f.names[i]; if (f.has_value) os << (f.names[i].size() == 1 ? "[" : "=[") << f.meta << "]"; }
		os << "\n                                        " << f.help << "\n";
	}
}

Args parse_cli(int argc, char** argv) {
	Args a;
	auto find = [&](const std::string& name) -> const FlagSpec* {
		for (const auto& f : FLAGS) for (const auto& n : f.names) if (n == name) return &f;
		return nullptr;
	};
	for (int i = 1; i < argc; ++i) {
		std::string tok = argv[i];
		std::string name, value;
		bool inline_value = false;
		if (tok.rfind("--", 0) == 0) {
			name = tok.substr(2);
			size_t eq = name.find('=');
			if (eq != std::string::npos) { value = name.substr(eq + 1); name = name.substr(0, eq); inline_value = true; }
		} else if (tok.size() >= 2 && tok[0] == '-') {
			name = tok.substr(1, 1);
			if (tok.size() > 2) { value = tok.substr(2); inline_value = true; }
		} else {
			throw ParseError("Passed in argument, but no positional arguments were ready to receive it: " + tok);
		}
		const FlagSpec* f = find(name);
		if (!f) throw ParseError("Flag could not be matched: " + name);
		if (f->has_value) {
			if (!inline_value) {
				if (i + 1 >= argc) throw ParseError("Flag '" + name + "' requires an argument but received none");
				value = argv[++i];
			}
		} else if (inline_value) {
			throw ParseError("Passed an argument into a non-argument flag: " + tok);
		}
		a.values[f->names.back()] = value;
	}
	return a;
}

#define RNB_CHECK(expr)                                                                          \
	do {                                                                                         \
		int rc_ = (expr);                                                                        \
		if (rc_ != RNB_OK) throw std::runtime_error(std::string(#expr) + ": " + rnb_last_error()); \
	} while (0)

// The parsed network configuration travels inside the snapshot (m_network_config, src/testbed.cu:3282-3313): JSON <-> MessagePack values.
static mpk::Value json_to_mpk(const jsonmin::Value& j) {
	switch (j.type) {
		case jsonmin::Value::Null: return mpk::Value();
		case jsonmin::Value::Bool: return mpk::Value::boolean(j.b);
		case jsonmin::Value::Number:
			if (j.num >= 0 && j.num == std::floor(j.num) && j.num < 1.8e19) return mpk::Value::uint((uint64_t)j.num);
			return mpk::Value::real(j.num);
		case jsonmin::Value::String: return mpk::Value::str(j.str);
		case jsonmin::Value::Arr: { mpk::Value a = mpk::Value::array(); for (const auto& e : *j.arr) a.arr.push_back(json_to_mpk(e)); return a; }
		default: { mpk::Value o = mpk::Value::object(); for (const auto& kv : *j.obj) o.set(kv.first, json_to_mpk(kv.second)); return o; }
	}
}
static jsonmin::Value mpk_to_json(const mpk::Value& m) {
	jsonmin::Value j;
	switch (m.type) {
		case mpk::Value::Nil: case mpk::Value::Bin: break;
		case mpk::Value::Bool: j.type = jsonmin::Value::Bool; j.b = m.b; break;
		case mpk::Value::Int: case mpk::Value::UInt: case mpk::Value::Float: j.type = jsonmin::Value::Number; j.num = m.number(); break;
		case mpk::Value::Str: j.type = jsonmin::Value::String; j.str = m.s; break;
		case mpk::Value::Arr: j.type = jsonmin::Value::Arr; j.arr = std::make_shared<jsonmin::Array>(); for (const auto& e : m.arr) j.arr->push_back(mpk_to_json(e)); break;
		default: j.type = jsonmin::Value::Obj; j.obj = std::make_shared<jsonmin::Object>(); for (const auto& kv : m.map) (*j.obj)[kv.first] = mpk_to_json(kv.second); break;
	}
	return j;
}

// ---- one process per GPU (the reference is single-GPU, so this has no counterpart in src/main.cu) ----
// Environment, set by tools/launch_testbed.sh (or torchrun-style variables): RNB_WORLD_SIZE | WORLD_SIZE, RNB_RANK | RANK,
// RNB_LOCAL_RANK | LOCAL_RANK (the HIP device), RNB_RCCL_ID_FILE (rank 0 writes three ncclUniqueIds there, the others wait for them).
// The job trains the SINGLE-GPU step: every rank takes 1/W of the rays and of the compacted batch (RNB_WEAK_SCALING=1: every
// rank keeps the configured sizes, the step grows W-fold). Per step: the 7 counters / loss sums are all-reduced on the
// library's device block, then the gradient blocks are exchanged in the order they become final through the SHARDED optimizer
// (the C++ form of dp.DataParallelTrainer._sharded_apply): reduce-scatter of a block -> Adam + EMA on this rank's 1/W of it ->
// all-gather of the fp16 training weights; block 0 (everything in front of the finest levels) on its own stream and its own
// channel beside the scatter of the finest levels. RNB_DP_SHARDED=0: all-reduce + replicated optimizer instead.
// The collectives go through a function table (dist_transport.hpp). Product: RCCL -- it orders the operations of ONE communicator, whatever
// streams they are given, so the three exchanges that are meant to run beside each other (step vector, early block, the rest) are three
// channels = three communicators, each with its own non-blocking stream. RNB_DP_TRANSPORT=staged + RNB_DP_STAGE_DIR (tests only) stages the
// same call sequence through host files, which is how the multi-rank path runs on two ranks sharing one GPU and in the CPU-checker build.
// Rank 0 alone writes meshes, snapshots and progress lines (after sync_parameters(): with the sharded optimizer a rank's fp32
// masters, EMA weights and Adam state are current on its own chunks only).
#ifdef RNB_WITH_RCCL
class RcclTransport : public dist::Transport {
	ncclComm_t comm_[3] = {nullptr, nullptr, nullptr};
	int world_;
	static void ok(ncclResult_t r, const char* what) { if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r)); }
	static ncclDataType_t type(dist::DType t) { return t == dist::F16 ? ncclHalf : t == dist::F64 ? ncclDouble : t == dist::U32 ? ncclUint32 : t == dist::I32 ? ncclInt32 : ncclFloat; }

public:
	RcclTransport(int world, int rank, const char* idf) : world_(world) {
		ncclUniqueId ids[3];
		if (world > 1 && !idf) throw std::runtime_error("RNB_RCCL_ID_FILE is not set (use tools/launch_testbed.sh)");
		if (rank == 0) {
			for (auto& id : ids) ok(ncclGetUniqueId(&id), "ncclGetUniqueId");
			if (idf) {
				const std::string tmp = std::string(idf) + ".tmp";
				std::FILE* f = std::fopen(tmp.c_str(), "wb");
				if (!f || std::fwrite(ids, sizeof(ids), 1, f) != 1) throw std::runtime_error("cannot write " + tmp);
				std::fclose(f);
				if (std::rename(tmp.c_str(), idf) != 0) throw std::runtime_error(std::string("cannot publish ") + idf);
			}
		} else {
			bool got = false;
			for (int tries = 0; tries < 1200 && !got; ++tries) { // up to two minutes
				if (std::FILE* f = std::fopen(idf, "rb")) { got = std::fread(ids, sizeof(ids), 1, f) == 1; std::fclose(f); }
				if (!got) usleep(100000);
			}
			if (!got) throw std::runtime_error(std::string("no ncclUniqueIds in ") + idf);
		}
		for (int k = 0; k < 3; ++k) ok(ncclCommInitRank(&comm_[k], world, ids[k], rank), "ncclCommInitRank");
		int n = 0;
		if (ncclCommCount(comm_[0], &n) != ncclSuccess || n != world) throw std::runtime_error("RCCL communicator does not span the job");
	}
	const char* name() const override { return "rccl"; }
	int n_ranks() const override { return world_; }
	void all_reduce(void* buf, size_t n, dist::DType t, dist::Op op, int chan, void* stream) override {
		ok(ncclAllReduce(buf, buf, n, type(t), op == dist::SUM ? ncclSum : ncclMax, comm_[chan], (hipStream_t)stream), "ncclAllReduce");
	}
	void reduce_scatter(void* block, void* own, size_t chunk, dist::DType t, int chan, void* stream) override {
		ok(ncclReduceScatter(block, own, chunk, type(t), ncclSum, comm_[chan], (hipStream_t)stream), "ncclReduceScatter");
	}
	void all_gather(const void* own, void* block, size_t chunk, dist::DType t, int chan, void* stream) override {
		ok(ncclAllGather(own, block, chunk, type(t), comm_[chan], (hipStream_t)stream), "ncclAllGather");
	}
	void shutdown() override {
		if (comm_[0]) (void)hipDeviceSynchronize();
		for (auto& c : comm_) if (c) { ncclCommDestroy(c); c = nullptr; }
	}
};
#endif

static std::string g_abort_file; // staged test transport: the file a failing rank leaves so that the others stop waiting for it

struct Dist {
	int world = 1, rank = 0, local_rank = 0;
	bool on = false, weak = false, sharded = true;
	std::unique_ptr<dist::Transport> tr;
	dist::StagedTransport* staged = nullptr; // (tr, when it is the test transport: a failing rank tells the others)
	enum { CH_MAIN = 0, CH_EARLY = 1, CH_VEC = 2 };
	// streams of the three channels; null in the CPU-checker build (no device: every call is synchronous there)
	void *s_main = nullptr, *s_early = nullptr, *s_vec = nullptr;
#ifdef RNB_WITH_HIP
	hipEvent_t ev_early = nullptr;
#endif
	double host7[7] = {0, 0, 0, 0, 0, 0, 0};
	double* host7_pinned = nullptr; // RCCL transport: the step vector's readback, pinned (an async copy on s_vec; the staged transport's synchronous helper is for the tests)
	bool synced = true; // no sharded update since the last sync_parameters()
	static int env_int(const char* a, const char* b, int def) {
		const char* v = std::getenv(a);
		if (!v) v = std::getenv(b);
		return v ? std::atoi(v) : def;
	}
	// how the staged transport and the step-vector readback reach a library buffer
#ifdef RNB_WITH_HIP
	static void mem_to_host(void* dst, const void* src, size_t bytes, void* stream) {
		if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || (bytes && hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess)) throw std::runtime_error("staged transport: device -> host copy failed");
	}
	static void mem_from_host(void* dst, const void* src, size_t bytes, void*) {
		if (bytes && hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("staged transport: host -> device copy failed");
	}
#else
	static void mem_to_host(void* dst, const void* src, size_t bytes, void*) { if (bytes) std::memcpy(dst, src, bytes); }
	static void mem_from_host(void* dst, const void* src, size_t bytes, void*) { if (bytes) std::memcpy(dst, src, bytes); }
#endif
	void init() {
		world = env_int("RNB_WORLD_SIZE", "WORLD_SIZE", 1);
		rank = env_int("RNB_RANK", "RANK", 0);
		local_rank = env_int("RNB_LOCAL_RANK", "LOCAL_RANK", rank);
		weak = std::getenv("RNB_WEAK_SCALING") != nullptr;
		if (const char* e = std::getenv("RNB_DP_SHARDED")) sharded = std::atoi(e) != 0;
		on = world > 1 || std::getenv("RNB_DP_FORCE_COLLECTIVES") != nullptr; // the variable exercises the collective path on one rank
		if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("bad RNB_WORLD_SIZE / RNB_RANK");
		if (!on) return;
		const char* which = std::getenv("RNB_DP_TRANSPORT");
		const bool want_staged = which && std::string(which) == "staged";
		if (which && !want_staged && std::string(which) != "rccl") throw std::runtime_error("RNB_DP_TRANSPORT must be rccl or staged");
#ifdef RNB_WITH_HIP
		if (hipSetDevice(local_rank) != hipSuccess) throw std::runtime_error("hipSetDevice(" + std::to_string(local_rank) + ") failed");
#endif
		if (want_staged) {
			const char* dir = std::getenv("RNB_DP_STAGE_DIR");
			if (!dir || !*dir) throw std::runtime_error("staged transport: RNB_DP_STAGE_DIR is not set");
			// every job in a directory of its own (tools/launch_testbed.sh exports a fresh RNB_DP_JOB_ID per launch): a directory reused after a crashed or killed run still
			// holds that run's last files and its `abort` marker, which the next job would fold into its sums or stop at
			std::string job_dir = dir;
			if (const char* job = std::getenv("RNB_DP_JOB_ID")) { ::mkdir(dir, 0777); job_dir += std::string("/job_") + job; }
			staged = new dist::StagedTransport(job_dir, world, rank, dist::MemOps{&Dist::mem_to_host, &Dist::mem_from_host});
			tr.reset(staged);
			g_abort_file = job_dir + "/abort";
		} else {
#ifdef RNB_WITH_RCCL
			tr.reset(new RcclTransport(world, rank, std::getenv("RNB_RCCL_ID_FILE")));
#else
			throw std::runtime_error("this build of testbed has no RCCL support (RNB_WORLD_SIZE > 1)");
#endif
		}
		if (rank == 0) std::cout << (want_staged ? "staged_ranks: " : "rccl_ranks: ") << tr->n_ranks() << (sharded ? " (sharded optimizer)" : " (all-reduce, replicated optimizer)") << std::endl;
#ifdef RNB_WITH_HIP
		hipStream_t a, b, c;
		if (hipStreamCreateWithFlags(&a, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&b, hipStreamNonBlocking) != hipSuccess ||
		    hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev_early, hipEventDisableTiming) != hipSuccess) throw std::runtime_error("stream creation failed");
		s_main = a; s_early = b; s_vec = c;
		if (!want_staged && hipHostMalloc((void**)&host7_pinned, 7 * sizeof(double), hipHostMallocDefault) != hipSuccess) throw std::runtime_error("hipHostMalloc failed");
#endif
	}
	// sizes of ONE rank (dp.strong_scaling_sizes of the Python side)
	void apply_sizes(rnb_config& cfg) const {
		cfg.world_size = (uint32_t)world; cfg.rank = (uint32_t)rank;
		if (world == 1 || weak) return;
		if (cfg.target_batch_size % (128u * world)) throw std::runtime_error("batch_size must be a multiple of 128 x world size");
		cfg.target_batch_size /= (uint32_t)world;
		cfg.max_rays_per_batch = std::max(128u, cfg.max_rays_per_batch / (uint32_t)world);
		cfg.initial_rays_per_batch = std::max(1u, cfg.initial_rays_per_batch / (uint32_t)world);
	}
	static char* buffer(rnb_ctx* ctx, int id) {
		void* p; uint64_t nb;
		if (rnb_buffer(ctx, id, &p, &nb) != RNB_OK) throw std::runtime_error(rnb_last_error());
		return (char*)p;
	}
	// the gradient vector of the context's accumulate mode: fp32 accumulators, or the half vector the ranks then sum in half
	bool half_grads = false;
	int grads_id() const { return half_grads ? RNB_BUF_GRADS_FP16 : RNB_BUF_GRADS_FP32; }
	dist::DType grads_type() const { return half_grads ? dist::F16 : dist::F32; }
	size_t grads_elem() const { return half_grads ? 2 : 4; }
	// the early stream's work (blocks exchanged beside the scatter) joins the main stream
	void join_early() {
#ifdef RNB_WITH_HIP
		if (hipEventRecord(ev_early, (hipStream_t)s_early) != hipSuccess || hipStreamWaitEvent((hipStream_t)s_main, ev_early, 0) != hipSuccess) throw std::runtime_error("early stream join failed");
#endif
	}
	// one block of the sharded optimizer on channel `chan` / stream `st`: reduce-scatter in place (the own chunk receives the sum), Adam + EMA on the own
	// chunk, all-gather of the fp16 training weights in place
	int shard_block(rnb_ctx* ctx, const rnb_shard_part& p, uint32_t k, char* g, char* w16, int chan, void* st) {
		const uint64_t chunk = p.own_hi - p.own_lo;
		int rc = rnb_gradient_part_wait(ctx, k, st);
		if (rc != RNB_OK) return rc;
		tr->reduce_scatter(g + p.lo * grads_elem(), g + p.own_lo * grads_elem(), chunk, grads_type(), chan, st);
		rc = rnb_train_step_apply_shard(ctx, k, st);
		if (rc != RNB_OK) return rc;
		tr->all_gather(w16 + p.own_lo * 2, w16 + p.lo * 2, chunk, dist::F16, chan, st);
		return RNB_OK;
	}
	// Occupancy updates sharded over the ranks (rnb_set_grid_exchange): the element-wise max of the splat targets, one all-reduce of 8 MB every 16 steps
	// instead of every rank evaluating all 2^20 samples. RNB_DP_SHARD_GRID=0 keeps the updates replicated.
	static int grid_exchange(void* user, void* grid_tmp, uint64_t n_elements, void* stream) {
		Dist* d = static_cast<Dist*>(user);
		try { d->tr->all_reduce(grid_tmp, n_elements, dist::U32, dist::MAX, CH_MAIN, stream); } // the order of the single-rank splat: atomicMax on the words as uint32 (a sign-bit NaN wins on both)
		catch (const std::exception& e) { std::cerr << "grid exchange: " << e.what() << std::endl; return -1; }
		return 0;
	}
	void attach(rnb_ctx* ctx, const rnb_config& cfg) {
		half_grads = cfg.accumulate == RNB_ACCUM_HALF;
		const char* e = std::getenv("RNB_DP_SHARD_GRID");
		if (on && !(e && std::atoi(e) == 0) && rnb_set_grid_exchange(ctx, &Dist::grid_exchange, this) != RNB_OK) throw std::runtime_error(rnb_last_error());
	}
	int train_step(rnb_ctx* ctx, rnb_step_stats* st) {
		if (!on) return rnb_train_step(ctx, nullptr, st);
		try { return train_step_collective(ctx, st); }
		catch (...) { if (staged) staged->abort_job(); throw; }
	}
	int train_step_collective(rnb_ctx* ctx, rnb_step_stats* st) {
		int rc = rnb_train_step_begin(ctx, s_main);
		if (rc != RNB_OK) return rc;
		uint64_t cnt[4]; double sums[3];
		rc = rnb_train_step_local(ctx, s_main, cnt, sums); // the host waits for the loss pass only
		if (rc != RNB_OK) return rc;
		char* vec = buffer(ctx, RNB_BUF_STEP_VECTOR);
		tr->all_reduce(vec, 7, dist::F64, dist::SUM, CH_VEC, s_vec);
#ifdef RNB_WITH_HIP
		if (host7_pinned) { // product path: asynchronous on the channel's own stream into pinned memory (a pageable hipMemcpy goes through the null stream and waits for every blocking stream of the process)
			if (hipMemcpyAsync(host7_pinned, vec, 7 * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)s_vec) != hipSuccess || hipStreamSynchronize((hipStream_t)s_vec) != hipSuccess) throw std::runtime_error("step vector readback failed");
			std::memcpy(host7, host7_pinned, sizeof(host7));
		} else
#endif
		mem_to_host(host7, vec, 7 * sizeof(double), s_vec);
		for (int k = 0; k < 4; ++k) cnt[k] = (uint64_t)std::llround(host7[k]);
		for (int k = 0; k < 3; ++k) sums[k] = host7[4 + k];
		const int rc_finish = rnb_train_step_finish(ctx, cnt, sums, st); // ray controller; queues the next step's march
		if (rc_finish != RNB_OK && rc_finish != RNB_ERR_NO_SAMPLES) return rc_finish;
		// gradients: block by block in completion order; the optimizer runs even when the step had no samples
		char* g = buffer(ctx, grads_id());
		if (sharded) {
			rnb_shard_part parts[RNB_MAX_SHARD_PARTS]; uint32_t n_parts = 0; uint64_t capacity = 0;
			rc = rnb_shard_layout(ctx, parts, &n_parts, &capacity);
			if (rc != RNB_OK) return rc;
			char* w16 = buffer(ctx, RNB_BUF_PARAMS_FP16);
			uint32_t first = 0;
			if (n_parts > 1) { // every block but the last on the early stream, each as soon as its levels are final: beside the scatter of the levels behind it
				for (uint32_t k = 0; k + 1 < n_parts; ++k) {
					rc = shard_block(ctx, parts[k], k, g, w16, CH_EARLY, s_early);
					if (rc != RNB_OK) return rc;
				}
				first = n_parts - 1;
			}
			for (uint32_t k = first; k < n_parts; ++k) {
				rc = shard_block(ctx, parts[k], k, g, w16, CH_MAIN, s_main);
				if (rc != RNB_OK) return rc;
			}
			if (first) join_early();
			rc = rnb_params_changed(ctx); // the training weights were written through a pointer: the kernels' weight images are stale
			if (rc != RNB_OK) return rc;
			rc = rnb_train_step_apply_done(ctx, s_main);
			if (rc != RNB_OK) return rc;
			synced = false;
			return rc_finish;
		}
		uint64_t ranges[3][2]; uint32_t n_parts = 0;
		rc = rnb_gradient_parts(ctx, ranges, &n_parts);
		if (rc != RNB_OK) return rc;
		const size_t ge = grads_elem();
		uint32_t first = 0;
		if (n_parts > 1) {
			rc = rnb_gradient_part_wait(ctx, 0, s_early);
			if (rc != RNB_OK) return rc;
			tr->all_reduce(g + ranges[0][0] * ge, ranges[0][1] - ranges[0][0], grads_type(), dist::SUM, CH_EARLY, s_early);
			rc = rnb_train_step_apply_early(ctx, s_early); // Adam on that block, beside the scatter of the finest levels and their exchange
			if (rc != RNB_OK) return rc;
			for (uint32_t k = 1; k + 1 < n_parts; ++k) { // the first half of the finest levels, beside the scatter of the second
				rc = rnb_gradient_part_wait(ctx, k, s_early);
				if (rc != RNB_OK) return rc;
				tr->all_reduce(g + ranges[k][0] * ge, ranges[k][1] - ranges[k][0], grads_type(), dist::SUM, CH_EARLY, s_early);
			}
			if (n_parts > 2) join_early();
			first = n_parts - 1;
		}
		for (uint32_t k = first; k < n_parts; ++k) {
			rc = rnb_gradient_part_wait(ctx, k, s_main);
			if (rc != RNB_OK) return rc;
			tr->all_reduce(g + ranges[k][0] * ge, ranges[k][1] - ranges[k][0], grads_type(), dist::SUM, CH_MAIN, s_main);
		}
		rc = rnb_train_step_apply(ctx, s_main);
		if (rc != RNB_OK) return rc;
		return rc_finish;
	}
	// Sharded optimizer: all-gather the per-rank chunks of the fp32 masters, EMA weights and Adam state so that every rank holds them
	// whole (dp.DataParallelTrainer.sync_parameters). A collective: every rank calls it, before rank 0 writes a mesh or a snapshot.
	void sync_parameters(rnb_ctx* ctx) {
		if (!on || !sharded || synced) return;
		try {
			rnb_shard_part parts[RNB_MAX_SHARD_PARTS]; uint32_t n_parts = 0; uint64_t capacity = 0;
			if (rnb_shard_layout(ctx, parts, &n_parts, &capacity) != RNB_OK) throw std::runtime_error(rnb_last_error());
#ifdef RNB_WITH_HIP
			if (hipDeviceSynchronize() != hipSuccess) throw std::runtime_error("hipDeviceSynchronize failed");
#endif
			const struct { int id; dist::DType type; size_t size; } bufs[] = {{RNB_BUF_PARAMS_FP32, dist::F32, 4}, {RNB_BUF_PARAMS_EMA, dist::F16, 2}, {RNB_BUF_ADAM_M, dist::F32, 4},
			                                                              {RNB_BUF_ADAM_V, dist::F32, 4}, {RNB_BUF_ADAM_STEPS, dist::I32, 4}};
			for (const auto& b : bufs) {
				char* base = buffer(ctx, b.id);
				for (uint32_t k = 0; k < n_parts; ++k) tr->all_gather(base + parts[k].own_lo * b.size, base + parts[k].lo * b.size, parts[k].own_hi - parts[k].own_lo, b.type, CH_MAIN, s_main);
			}
#ifdef RNB_WITH_HIP
			if (hipStreamSynchronize((hipStream_t)s_main) != hipSuccess) throw std::runtime_error("sync_parameters failed");
#endif
			synced = true;
		} catch (...) { if (staged) staged->abort_job(); throw; }
	}
	void abort_job() { if (staged) staged->abort_job(); }
	void shutdown() { if (tr) tr->shutdown(); }
};

struct Testbed {
	rnb_config cfg;
	rnb_ctx* ctx = nullptr;
	Dataset ds;
	std::string scene, output_path;
	uint32_t max_iter = 15000; // testbed.h:503
	float loss_scalar = 0.f;
	bool train = true;
	bool fractional_training = false;
	bool accumulate_from_flag = false; // --accumulate was given: it overrides the mode a snapshot records
	uint32_t fractional = 0;
	uint32_t save_each = 0;
	uint32_t res_mesh = 512;
	std::string mesh_prefix;
	jsonmin::Value network_config;
	Dist dist;

	~Testbed() { if (ctx) rnb_destroy(ctx); }

	// What of a network config this build cannot honour is refused by name instead of being ignored. The architecture of the path is the reference's base.json
	// (nerf_network.h:40-83): density MLP 32 -> 64 -> 16, colour MLP 48 -> 64 -> 64 -> 16, two features per level.
	static void check_supported(const jsonmin::Value& c) {
		auto need = [](const jsonmin::Value& blk, const char* block, const char* key, double want) {
			if (!blk.contains(key) || blk[key].is_null()) return;
			const double v = blk[key].as_number();
			if (v != want) throw std::runtime_error(std::string("network config: ") + block + "." + key + " = " + std::to_string(v) + " is not supported by this build (fixed at " + std::to_string(want) + ")");
		};
		auto need_str = [](const jsonmin::Value& blk, const char* block, const char* key, const char* want) {
			if (!blk.contains(key) || blk[key].is_null()) return;
			if (blk[key].as_string() != want) throw std::runtime_error(std::string("network config: ") + block + "." + key + " = \"" + blk[key].as_string() + "\" is not supported by this build (fixed at \"" + want + "\")");
		};
		const auto& enc = c["encoding"];
		need(enc, "encoding", "n_features_per_level", 2);
		const uint32_t L = enc.value("n_levels", 16u);
		// NerfNetwork pads the density network's input [x y z | 2 L features] to a multiple of 16 and picks the geometric initialisation by that width
		// (load_sdf_mlp_weight, nerf_network.h:585-604): 32 -> utils/mlp_weights_hidden_layer_num_1_hidden_size_32.txt, 48 -> utils/mlp_weights.txt
		if (3 + 2 * L > 32) throw std::runtime_error("network config: encoding.n_levels = " + std::to_string(L) + " gives a density-network input of width " + std::to_string((3 + 2 * L + 15) / 16 * 16) +
		                                             " (the reference's utils/mlp_weights.txt case, nerf_network.h:595-600); this build supports width 32 only (n_levels <= 14)");
		for (const char* blk : {"network", "rgb_network"}) {
			need(c[blk], blk, "n_neurons", 64);
			need(c[blk], blk, "n_hidden_layers", std::string(blk) == "network" ? 1 : 2);
			need_str(c[blk], blk, "activation", "ReLU");
			need_str(c[blk], blk, "output_activation", "None");
		}
		need_str(c["optimizer"], "optimizer", "otype", "Ema");
	}

	void apply_network_config(const jsonmin::Value& c) { // Testbed::reset_network, src/testbed.cu:2245-2335
		check_supported(c);
		const auto& enc = c["encoding"];
		cfg.n_levels = enc.value("n_levels", 16u);
		cfg.log2_hashmap_size = enc.value("log2_hashmap_size", 15u);
		cfg.base_resolution = enc.value("base_resolution", 0u);
		if (!cfg.base_resolution) cfg.base_resolution = 1u << (cfg.log2_hashmap_size / 3);
		const float desired_resolution = enc.value("top_resolution", 2048.0f);
		float pls = enc.value("per_level_scale", 0.0f);
		if (pls <= 0.0f && cfg.n_levels > 1) pls = std::exp(std::log(desired_resolution * (float)ds.aabb_scale / (float)cfg.base_resolution) / (cfg.n_levels - 1));
		cfg.per_level_scale = pls;
		cfg.valid_level_scale = enc.value("valid_level_scale", 0.01f);
		cfg.base_valid_level_scale = enc.value("base_valid_level_scale", 0.5f);
		cfg.base_training_step = enc.value("base_training_step", 200u);
		cfg.sdf_bias = c["network"].value("sdf_bias", -0.1f);
		const auto& hp = c["hyperparams"];
		cfg.target_batch_size = hp.value("batch_size", 1u << 18);
		cfg.mask_loss_weight = hp.value("mask_loss_weight", 0.f);
		cfg.ek_loss_weight = hp.value("ek_loss_weight", 0.01f);
		const jsonmin::Value* opt = &c["optimizer"]; // Ema -> ExponentialDecay -> Adam
		if (opt->contains("decay")) cfg.ema_decay = (*opt)["decay"].as_float();
		if (opt->contains("nested")) {
			opt = &(*opt)["nested"];
			cfg.lr_decay_start = opt->value("decay_start", cfg.lr_decay_start);
			cfg.lr_decay_interval = opt->value("decay_interval", cfg.lr_decay_interval);
			cfg.lr_decay_base = opt->value("decay_base", cfg.lr_decay_base);
			if (opt->contains("nested")) {
				opt = &(*opt)["nested"];
				cfg.learning_rate = opt->value("learning_rate", cfg.learning_rate);
				cfg.beta1 = opt->value("beta1", cfg.beta1); cfg.beta2 = opt->value("beta2", cfg.beta2);
				cfg.epsilon = opt->value("epsilon", cfg.epsilon); cfg.l2_reg = opt->value("l2_reg", cfg.l2_reg);
			}
		}
		cfg.aabb_scale = (uint32_t)ds.aabb_scale;
	}

	void create_context() {
		if (ctx) { rnb_destroy(ctx); ctx = nullptr; }
		dist.apply_sizes(cfg);
		RNB_CHECK(rnb_create(&cfg, &ctx));
		dist.attach(ctx, cfg);
		// geometric initialisation of the SDF MLP (nerf_network.h:585-623): <exe_dir>/../utils/...
		const std::string wpath = parent_path(exe_dir()) + "/utils/mlp_weights_hidden_layer_num_1_hidden_size_32.txt";
		std::FILE* fp = std::fopen(wpath.c_str(), "r");
		std::printf("network_params_elements: %d\n", RNB_N_SDF_MLP_PARAMS);
		if (!fp) {
			std::printf("[ERROR] Load SDF MLP weight failed!\n[ERROR] Tried to load from: %s\n[ERROR] Please ensure the utils directory exists in the project root!\n", wpath.c_str());
			std::exit(1);
		}
		std::vector<float> w(RNB_N_SDF_MLP_PARAMS, 0.f);
		for (auto& v : w) if (std::fscanf(fp, "%f", &v) != 1) break;
		std::fclose(fp);
		RNB_CHECK(rnb_init_params(ctx, w.data()));
		std::vector<const uint16_t*> nm(ds.views.size()), al(ds.views.size());
		for (size_t i = 0; i < ds.views.size(); ++i) { nm[i] = ds.normals[i].rgba.data(); al[i] = ds.albedos[i].rgba.data(); }
		RNB_CHECK(rnb_set_dataset(ctx, (uint32_t)ds.views.size(), ds.views.data(), nm.data(), al.data()));
	}

	void load_training_data(const std::string& data_path) { // Testbed::load_training_data, src/testbed.cu:83-122
		scene = data_path;
		output_path = data_path + "/output";
		make_dir(output_path);
		if (std::FILE* lf = std::fopen((output_path + "/log.txt").c_str(), "wb")) std::fclose(lf);
		make_dir(output_path + "/mesh");
		make_dir(output_path + "/images");
		const auto t0 = std::chrono::steady_clock::now();
		ds = load_dataset(data_path);
		std::printf("Loaded %zu images after %.1fs\n", ds.views.size(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
	}

	// ---- snapshot (src/testbed.cu:3280-3390; tiny-cuda-nn/trainer.h:263-305) ----
	template <typename T> std::vector<T> download(int buf) {
		void* p; uint64_t nb;
		RNB_CHECK(rnb_buffer(ctx, buf | RNB_BUF_READONLY, &p, &nb));
		std::vector<T> h(nb / sizeof(T));
		if (nb) RNB_CHECK(rnb_memcpy(ctx, h.data(), p, nb, RNB_D2H));
		return h;
	}
	static uint16_t f32_to_f16(float f) {
		uint32_t x; std::memcpy(&x, &f, 4);
		const uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
		if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u : 0));
		if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
		if (ax < 0x33000001u) return (uint16_t)sign;
		const int e = (int)(ax >> 23) - 127;
		const uint32_t m = (ax & 0x7fffffu) | 0x800000u;
		const int shift = e < -14 ? 13 + (-14 - e) : 13;
		const uint32_t hexp = e < -14 ? 0 : (uint32_t)(e + 15);
		uint32_t hm = m >> shift;
		const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
		if (rem > half || (rem == half && (hm & 1u))) ++hm;
		return (uint16_t)(sign | (hexp == 0 ? hm : ((hexp - 1) << 10) + hm));
	}
	static float f16_to_f32(uint16_t h) {
		const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu;
		uint32_t man = h & 0x3ffu, bits;
		if (exp == 0) {
			if (man == 0) bits = sign;
			else { int e = -1; do { ++e; man <<= 1; } while ((man & 0x400u) == 0); bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13; }
		} else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
		else bits = sign | (exp + 127 - 15) << 23 | man << 13;
		float f; std::memcpy(&f, &bits, 4); return f;
	}

	void save_snapshot(const std::string& path, uint32_t training_step, uint32_t rays_per_batch, uint32_t measured, uint32_t measured_before) {
		mpk::Value root = mpk::Value::object();
		mpk::Value snap = mpk::Value::object();
		const auto ema = download<uint16_t>(RNB_BUF_PARAMS_EMA); // Trainer::serialize: the inference (EMA) weights, fp16
		snap.set("n_params", mpk::Value::uint(ema.size()));
		snap.set("params_binary", mpk::Value::binary(ema.data(), ema.size() * 2));
		snap.set("density_grid_size", mpk::Value::uint(RNB_GRIDSIZE));
		const auto grid = download<float>(RNB_BUF_DENSITY_GRID);
		std::vector<uint16_t> g16(grid.size());
		for (size_t i = 0; i < grid.size(); ++i) g16[i] = f32_to_f16(grid[i]);
		snap.set("density_grid_binary", mpk::Value::binary(g16.data(), g16.size() * 2));
		mpk::Value nerf = mpk::Value::object();
		nerf.set("aabb_scale", mpk::Value::uint((uint64_t)ds.aabb_scale));
		mpk::Value rgb = mpk::Value::object();
		rgb.set("rays_per_batch", mpk::Value::uint(rays_per_batch));
		rgb.set("measured_batch_size", mpk::Value::uint(measured));
		rgb.set("measured_batch_size_before_compaction", mpk::Value::uint(measured_before));
		nerf.set("rgb", rgb);
		snap.set("nerf", nerf);
		snap.set("training_step", mpk::Value::uint(training_step));
		snap.set("loss", mpk::Value::real(loss_scalar));
		{ // The reference's loader also reads the accumulated global movement and the delta network (load_global_movement / load_local_movement,
			// nerf_network.h:1017-1081): half buffers, identity here (static scenes; nerf_network.h:852-905, transform_network.h:30-38, 209-235).
			const uint16_t one = f32_to_f16(1.0f);
			const uint16_t rot[12] = {one, 0, 0, 0, one, 0, 0, 0, one, 0, 0, 0}, tr[4] = {0, 0, 0, 0}, lrot[8] = {one, 0, 0, 0, one, 0, 0, 0};
			snap.set("rotation", mpk::Value::binary(rot, sizeof(rot)));
			snap.set("transition", mpk::Value::binary(tr, sizeof(tr)));
			snap.set("local_rotation", mpk::Value::binary(lrot, sizeof(lrot)));
			snap.set("local_transition", mpk::Value::binary(tr, sizeof(tr)));
		}
		root.set("snapshot", snap);
		// the whole network configuration travels with the snapshot (m_network_config, src/testbed.cu:3282-3313): optimizer, loss, both
		// encodings and networks as parsed, so that a resumed run steps with the hyper-parameters of the run that wrote the file
		if (network_config.is_object())
			for (const auto& kv : *network_config.obj) if (kv.first != "snapshot") root.set(kv.first, json_to_mpk(kv.second));
		auto obj = [&](const char* k) -> mpk::Value& { // the block named k, created if the config had none
			for (auto& kv : root.map) if (kv.first == k && kv.second.type == mpk::Value::MapT) return kv.second;
			return root.set(k, mpk::Value::object());
		};
		// ... overlaid with the values in effect (derived resolutions, command-line overrides such as --mask-weight)
		mpk::Value& enc = obj("encoding");
		if (!enc.find("otype")) enc.set("otype", mpk::Value::str("HashGrid"));
		enc.set("n_levels", mpk::Value::uint(cfg.n_levels)); enc.set("n_features_per_level", mpk::Value::uint(2));
		enc.set("log2_hashmap_size", mpk::Value::uint(cfg.log2_hashmap_size)); enc.set("base_resolution", mpk::Value::uint(cfg.base_resolution));
		enc.set("per_level_scale", mpk::Value::real(cfg.per_level_scale));
		enc.set("valid_level_scale", mpk::Value::real(cfg.valid_level_scale)); enc.set("base_valid_level_scale", mpk::Value::real(cfg.base_valid_level_scale));
		enc.set("base_training_step", mpk::Value::uint(cfg.base_training_step));
		mpk::Value& hp = obj("hyperparams");
		hp.set("batch_size", mpk::Value::uint((uint64_t)cfg.target_batch_size * ((dist.world > 1 && !dist.weak) ? dist.world : 1))); // the job's batch, not this rank's share
		hp.set("mask_loss_weight", mpk::Value::real(cfg.mask_loss_weight)); hp.set("ek_loss_weight", mpk::Value::real(cfg.ek_loss_weight));
		hp.set("accumulate", mpk::Value::str(cfg.accumulate == RNB_ACCUM_HALF ? "half" : "fp32")); // (this build's key: a resumed run continues in the mode the snapshot was trained in)
		hp.set("deterministic", mpk::Value::boolean(cfg.deterministic != 0));
		mpk::Value& net = obj("network");
		if (!net.find("otype")) net.set("otype", mpk::Value::str("FullyFusedMLP"));
		net.set("sdf_bias", mpk::Value::real(cfg.sdf_bias));
		{ // Ema -> ExponentialDecay -> Adam, the nesting of configs/nerf/base.json
			mpk::Value& ema = obj("optimizer");
			if (!ema.find("otype")) ema.set("otype", mpk::Value::str("Ema"));
			ema.set("decay", mpk::Value::real(cfg.ema_decay));
			auto nested = [](mpk::Value& parent) -> mpk::Value& {
				for (auto& kv : parent.map) if (kv.first == "nested" && kv.second.type == mpk::Value::MapT) return kv.second;
				return parent.set("nested", mpk::Value::object());
			};
			mpk::Value& dec = nested(ema);
			if (!dec.find("otype")) dec.set("otype", mpk::Value::str("ExponentialDecay"));
			dec.set("decay_start", mpk::Value::uint(cfg.lr_decay_start)); dec.set("decay_interval", mpk::Value::uint(cfg.lr_decay_interval));
			dec.set("decay_base", mpk::Value::real(cfg.lr_decay_base));
			mpk::Value& adam = nested(dec);
			if (!adam.find("otype")) adam.set("otype", mpk::Value::str("Adam"));
			adam.set("learning_rate", mpk::Value::real(cfg.learning_rate)); adam.set("beta1", mpk::Value::real(cfg.beta1)); adam.set("beta2", mpk::Value::real(cfg.beta2));
			adam.set("epsilon", mpk::Value::real(cfg.epsilon)); adam.set("l2_reg", mpk::Value::real(cfg.l2_reg));
		}
		mpk::save(path, root);
	}

	struct Resume { uint32_t training_step = 0, rays_per_batch = 0, measured_before = 0; };
	Resume load_snapshot(const std::string& path) { // Testbed::load_snapshot, src/testbed.cu:3333-3390
		const mpk::Value root = mpk::load(path);
		if (!root.find("snapshot")) throw std::runtime_error("File '" + path + "' does not contain a snapshot.");
		const mpk::Value& snap = root.at("snapshot");
		if ((uint32_t)snap.at("density_grid_size").number() != RNB_GRIDSIZE) throw std::runtime_error("Incompatible grid size.");
		{ // reset_network from the snapshot's own config (src/testbed.cu:3352-3357): every block but the binary payload
			mpk::Value cfg_root = mpk::Value::object();
			for (const auto& kv : root.map) if (kv.first != "snapshot") cfg_root.set(kv.first, kv.second);
			network_config = mpk_to_json(cfg_root);
			const float mask_w = cfg.mask_loss_weight;
			apply_network_config(network_config);
			if (!network_config["hyperparams"].contains("mask_loss_weight")) cfg.mask_loss_weight = mask_w;
			if (!accumulate_from_flag && network_config["hyperparams"].contains("accumulate")) cfg.accumulate = network_config["hyperparams"]["accumulate"].as_string() == "half" ? RNB_ACCUM_HALF : RNB_ACCUM_FP32;
		}
		if (const mpk::Value* v = snap.at("nerf").find("aabb_scale")) cfg.aabb_scale = (uint32_t)v->number();
		create_context();
		const auto& pb = snap.at("params_binary").bin;
		const uint64_t n = rnb_n_params(ctx);
		if (pb.size() != n * 2) throw std::runtime_error("Can't set params because CPU buffer has the wrong size.");
		std::vector<float> p32(n);
		const uint16_t* ph = reinterpret_cast<const uint16_t*>(pb.data());
		for (uint64_t i = 0; i < n; ++i) p32[i] = f16_to_f32(ph[i]); // Trainer::set_params: master = float(half)
		RNB_CHECK(rnb_set_params(ctx, p32.data()));
		const auto& gb = snap.at("density_grid_binary").bin;
		void* gp; uint64_t gnb;
		RNB_CHECK(rnb_buffer(ctx, RNB_BUF_DENSITY_GRID, &gp, &gnb));
		if (gb.size() / 2 == gnb / 4) {
			std::vector<float> g32(gb.size() / 2);
			const uint16_t* gh = reinterpret_cast<const uint16_t*>(gb.data());
			for (size_t i = 0; i < g32.size(); ++i) g32[i] = f16_to_f32(gh[i]);
			RNB_CHECK(rnb_memcpy(ctx, gp, g32.data(), gnb, RNB_H2D));
			RNB_CHECK(rnb_update_density_bitfield(ctx, nullptr));
		} else if (!gb.empty()) throw std::runtime_error("Incompatible number of grid cascades.");
		Resume r;
		r.training_step = (uint32_t)snap.at("training_step").number();
		loss_scalar = (float)snap.at("loss").number();
		const mpk::Value& rgb = snap.at("nerf").at("rgb");
		r.rays_per_batch = (uint32_t)rgb.at("rays_per_batch").number();
		r.measured_before = (uint32_t)rgb.at("measured_batch_size_before_compaction").number();
		RNB_CHECK(rnb_set_controller(ctx, r.training_step, std::max(1u, std::min(r.rays_per_batch, cfg.max_rays_per_batch)), r.measured_before, 0));
		return r;
	}

	// ---- mesh (src/testbed_nerf.cu:4218-4350, src/marching_cubes.cu) ----
	void compute_and_save_marching_cubes_mesh(const std::string& filename, uint32_t res_in) {
		const uint32_t res = (res_in + 15u) / 16u * 16u; // next_multiple(res, 16)
		std::printf("unwrap_it:0\n%u%u%u\n", res, res, res);
		const float amin = 0.5f - 0.5f * (float)cfg.aabb_scale, amax = 0.5f + 0.5f * (float)cfg.aabb_scale;
		const float mn[3] = {amin, amin, amin}, mx[3] = {amax, amax, amax};
		const size_t n = (size_t)res * res * res;
		const float diag = amax - amin;
		// get_density_on_grid + marching_cubes_gpu on the device (src/testbed_nerf.cu:4218-4269, src/marching_cubes.cu:794-822): the
		// lattice (4 bytes per point) and the edge -> vertex grid (12 bytes per point) never leave it; EMA weights
		const uint32_t res3[3] = {res, res, res};
		void* lattice = nullptr;
		RNB_CHECK(rnb_device_malloc(ctx, (uint64_t)n * 4, &lattice));
		RNB_CHECK(rnb_sdf_lattice(ctx, nullptr, res3, amin, amax, (float*)lattice, 1));
		float* dverts = nullptr;
		uint32_t* dindices = nullptr;
		uint32_t nv = 0, ni = 0;
		RNB_CHECK(rnb_marching_cubes(ctx, nullptr, (const float*)lattice, res3, mn, mx, 0.0f, &dverts, &dindices, &nv, &ni));
		rnb_device_free(ctx, lattice);
		mesh::Mesh m;
		m.verts.resize(nv); m.indices.resize(ni);
		if (nv) RNB_CHECK(rnb_memcpy(ctx, m.verts.data(), dverts, (uint64_t)nv * 12, RNB_D2H));
		if (ni) RNB_CHECK(rnb_memcpy(ctx, m.indices.data(), dindices, (uint64_t)ni * 4, RNB_D2H));
		rnb_device_free(ctx, dverts); rnb_device_free(ctx, dindices);
		mesh::compute_normals(m);
		std::printf("#vertices=%zu #triangles=%zu\n", m.verts.size(), m.indices.size() / 3);
		// vertex colours: full network at the vertices, outward view direction (src/testbed_nerf.cu:793-799, 4193-4216)
		m.colors.assign(m.verts.size(), {0, 0, 0});
		if (!m.verts.empty()) {
			void *dc = nullptr, *dq = nullptr;
			const uint32_t cb = 1u << 18;
			RNB_CHECK(rnb_device_malloc(ctx, (uint64_t)cb * 28, &dc));
			RNB_CHECK(rnb_device_malloc(ctx, (uint64_t)cb * 32, &dq));
			std::vector<float> hc((size_t)cb * 7);
			std::vector<uint16_t> ho((size_t)cb * 16);
			for (size_t off = 0; off < m.verts.size(); off += cb) {
				const uint32_t nb = (uint32_t)std::min<size_t>(cb, m.verts.size() - off);
				for (uint32_t q = 0; q < nb; ++q) {
					const mesh::Vec3 v = m.verts[off + q];
					float d[3] = {v.x - 0.5f, v.y - 0.5f, v.z - 0.5f};
					const float l = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
					float* c = &hc[(size_t)q * 7];
					c[0] = (v.x - amin) / diag; c[1] = (v.y - amin) / diag; c[2] = (v.z - amin) / diag; c[3] = 0.f;
					for (int k = 0; k < 3; ++k) c[4 + k] = (d[k] / l + 1.f) * 0.5f;
				}
				RNB_CHECK(rnb_memcpy(ctx, dc, hc.data(), (uint64_t)nb * 28, RNB_H2D));
				RNB_CHECK(rnb_forward_infer(ctx, nullptr, (const float*)dc, nb, (uint16_t*)dq, 1));
				RNB_CHECK(rnb_memcpy(ctx, ho.data(), dq, (uint64_t)nb * 32, RNB_D2H));
				for (uint32_t q = 0; q < nb; ++q) {
					auto sig = [&](int k) { return 1.f / (1.f + std::exp(-f16_to_f32(ho[(size_t)q * 16 + k]))); }; // rgb_activation = Logistic
					m.colors[off + q] = {sig(0), sig(1), sig(2)};
				}
			}
			rnb_device_free(ctx, dc); rnb_device_free(ctx, dq);
		}
		mesh::save_obj(filename, m, ds.scale, ds.offset, ds.n2w_s, ds.n2w_t, ds.from_na);
	}
};

} // namespace

int main(int argc, char** argv) {
	Args args;
	try {
		args = parse_cli(argc, argv);
	} catch (const ParseError& e) {
		std::cerr << e.what() << std::endl;
		print_help(std::cerr, argv[0]);
		return -1;
	} catch (const ValidationError& e) {
		std::cerr << e.what() << std::endl;
		print_help(std::cerr, argv[0]);
		return -2;
	}
	if (args.has("help")) { print_help(std::cout, argv[0]); return 0; }
	if (args.has("version")) { std::cout << "neural graphics primitives version " NGP_VERSION << std::endl; return 0; }
	try {
		Testbed tb;
		rnb_default_config(&tb.cfg);
		if (args.has("accumulate")) { // (not a flag of the reference: its arithmetic is the `half` mode, include/rnb_neus2.h rnb_config::accumulate)
			const std::string m = args.get("accumulate");
			if (m != "fp32" && m != "half") { std::cerr << "--accumulate takes fp32 or half" << std::endl; print_help(std::cerr, argv[0]); return -1; }
			tb.cfg.accumulate = m == "half" ? RNB_ACCUM_HALF : RNB_ACCUM_FP32;
			tb.accumulate_from_flag = true;
		}
		if (args.has("deterministic")) tb.cfg.deterministic = 1; // (not a flag of the reference, whose runs are not reproducible: include/rnb_neus2.h rnb_config::deterministic)
		tb.dist.init();
		struct AbortGuard { bool done = false; ~AbortGuard() { if (!done && !g_abort_file.empty()) if (std::FILE* f = std::fopen(g_abort_file.c_str(), "wb")) std::fclose(f); } } abort_guard; // any exit but the regular one
		const bool lead = tb.dist.rank == 0; // meshes, snapshots and progress lines come from rank 0 only
		try {
			if (args.has("maxiter")) tb.max_iter = args.get_u32("maxiter");
			if (args.has("resolution")) tb.res_mesh = args.get_u32("resolution");
			if (args.has("save-each")) tb.save_each = args.get_u32("save-each");
			if (args.has("fractional-training")) (void)args.get_u32("fractional-training");
			if (args.has("mask-weight")) (void)args.get_f32("mask-weight");
			if (args.has("width")) (void)args.get_u32("width");
			if (args.has("height")) (void)args.get_u32("height");
		} catch (const ParseError& e) {
			std::cerr << e.what() << std::endl;
			print_help(std::cerr, argv[0]);
			return -1;
		}
		std::cout << "Number of iterations : " << tb.max_iter << std::endl;
		if (args.has("scene")) {
			std::string scene_path = args.get("scene");
			if (!path_exists(scene_path)) { std::cerr << "Scene path " << scene_path << " does not exist." << std::endl; return 1; }
			while (scene_path.size() > 1 && scene_path.back() == '/') scene_path.pop_back();
			tb.load_training_data(scene_path);
		} else {
			std::cerr << "No scene given (--scene)." << std::endl; // the reference dereferences the missing flag (src/main.cu:413)
			return 1;
		}
		Testbed::Resume resume;
		if (args.has("snapshot")) {
			const std::string sp = args.get("snapshot");
			if (!path_exists(sp)) { std::cerr << "Snapshot path " << sp << " does not exist." << std::endl; return 1; }
			resume = tb.load_snapshot(sp);
			tb.train = true;
			std::printf("*******Loaded snapshot succeed!\n");
		} else {
			std::string cfg_path = parent_path(exe_dir()) + "/configs/nerf";
			if (args.has("config")) {
				const std::string c = args.get("config");
				cfg_path = path_exists(cfg_path + "/" + c) ? cfg_path + "/" + c : c;
			} else cfg_path += "/base.json";
			if (!path_exists(cfg_path)) { std::cerr << "Network config path " << cfg_path << " does not exist." << std::endl; return 1; }
			std::cout << "Network config path " << cfg_path << " found!" << std::endl;
			tb.network_config = jsonmin::parse_file(cfg_path);
			tb.apply_network_config(tb.network_config);
			tb.create_context();
			tb.train = !args.has("no-train");
		}
		// flag setters (src/main.cu:349-410)
		if (args.has("mask-weight")) tb.cfg.mask_loss_weight = args.get_f32("mask-weight");
		if (args.has("fractional-training")) {
			if (args.has("maxiter")) {
				const uint32_t v = args.get_u32("fractional-training");
				if (v < tb.max_iter) { tb.fractional = v; tb.fractional_training = true; }
				else { std::cerr << "The integer must be lower than max-iter!" << std::endl; return 1; }
			} else { std::cerr << "fractional-training works with max-iter." << std::endl; return 1; }
		}
		tb.cfg.apply_L2 = args.has("lone") ? 0 : 1;
		tb.cfg.apply_supernormal = args.has("supernormal") ? 1 : 0;
		tb.cfg.apply_rgbplus = args.has("no-rgbplus") ? 0 : 1;
		if (args.has("disable-snap-to-center")) tb.cfg.snap_to_pixel_centers = 0;
		tb.cfg.apply_bce = args.has("bce") ? 1 : 0;
		tb.cfg.apply_relu = args.has("relu") ? 1 : 0;
		tb.cfg.apply_light_opti = args.has("opti-lights") ? 1 : 0;
		tb.cfg.apply_no_albedo = args.has("no-albedo") ? 1 : 0;
		RNB_CHECK(rnb_update_config(tb.ctx, &tb.cfg));
		const std::string obj_filename = tb.output_path + "/mesh_" + std::to_string(tb.max_iter) + ".obj";
		tb.mesh_prefix = tb.output_path + "/mesh_";

		// training loop (src/main.cu:444-453, Testbed::frame src/testbed.cu:1826-1919)
		uint32_t step = rnb_training_step(tb.ctx);
		uint32_t rays_per_batch = rnb_rays_per_batch(tb.ctx), measured = 0, measured_before = resume.measured_before;
		double train_ms = 0.0; uint64_t rays_total = 0;
		const bool no_train = args.has("no-train");
		if (!no_train) {
			bool running = true;
			while (running) {
				if (tb.train) {
					rnb_step_stats st;
					const int rc = tb.dist.train_step(tb.ctx, &st);
					if (rc == RNB_ERR_NO_SAMPLES) { std::cout << "Nerf training generated 0 samples. Aborting training." << std::endl; tb.train = false; tb.loss_scalar = 0.f; }
					else if (rc != RNB_OK) throw std::runtime_error(rnb_last_error());
					else {
						tb.loss_scalar = st.loss; // m_loss_scalar.val() is the latest step's loss (common.h:264-281)
						train_ms += st.prep_ms + st.step_ms; rays_total += st.rays_per_batch;
						rays_per_batch = st.next_rays_per_batch; measured = st.measured_batch_size; measured_before = st.measured_batch_size_before_compaction;
					}
					step = rnb_training_step(tb.ctx);
				}
				if (tb.fractional_training) { // src/testbed.cu:1886-1895
					const bool sdf_only = step < tb.fractional;
					if ((tb.cfg.apply_no_albedo != 0) != sdf_only || (tb.cfg.only_sdf_training != 0) != sdf_only) {
						tb.cfg.apply_no_albedo = sdf_only; tb.cfg.only_sdf_training = sdf_only;
						RNB_CHECK(rnb_update_config(tb.ctx, &tb.cfg));
					}
				}
				if (tb.save_each > 0 && step % tb.save_each == 0) tb.dist.sync_parameters(tb.ctx); // collective: every rank
				if (lead && tb.save_each > 0 && step % tb.save_each == 0) {
					const std::string name = tb.mesh_prefix + std::to_string(step) + ".obj";
					std::printf("%s\n", name.c_str());
					tb.compute_and_save_marching_cubes_mesh(name, tb.res_mesh);
				}
				running = step < tb.max_iter; // Testbed::frame's return value
				if (lead && running && step % 100 == 0) std::cout << "iteration=" << step << " loss=" << tb.loss_scalar << std::endl; // src/main.cu:444-451
				if (!tb.train) running = false; // the reference would spin here forever once training aborted; stop instead
			}
			if (lead && train_ms > 0) std::cout << "throughput: " << (double)rays_total * tb.dist.world / (train_ms * 1e-3) << " rays/s, " << train_ms / std::max(1u, step - resume.training_step) << " ms/step" << std::endl;
		}
		if (args.has("save-mesh") || args.has("save-snapshot")) tb.dist.sync_parameters(tb.ctx); // collective: every rank
		if (lead && args.has("save-mesh")) {
			// --free-memory releases the dataset before meshing in the reference (src/main.cu:455-459; 10 s sleep not reproduced)
			tb.compute_and_save_marching_cubes_mesh(obj_filename, tb.res_mesh);
		}
		const std::string snapshot_filename = tb.output_path + "/snapshot_" + std::to_string(tb.max_iter) + ".msgpack";
		if (lead && args.has("save-snapshot")) {
			std::cout << "Saving Snapshot !" << std::endl << snapshot_filename << std::endl;
			tb.save_snapshot(snapshot_filename, step, rays_per_batch, measured, measured_before);
		}
		tb.dist.shutdown();
		abort_guard.done = true;
	} catch (const std::exception& e) {
		std::cerr << "Uncaught exception: " << e.what() << std::endl;
		if (!g_abort_file.empty()) if (std::FILE* f = std::fopen(g_abort_file.c_str(), "wb")) std::fclose(f);
		return 1;
	}
	return 0;
}
