// dist_transport.hpp — the collectives of the one-process-per-GPU `testbed` (struct Dist, testbed_main.cpp) behind a function table.
//
// The reference is single-GPU (the partitioning rests on src/testbed_nerf.cu:1213 alone), so none of this has a counterpart there. Dist uses five
// collectives: a sum all-reduce of the 7-value step vector, reduce-scatter / all-gather of parameter-shaped blocks (sharded optimizer), a sum all-reduce of gradient
// blocks (replicated optimizer) and a max all-reduce of the occupancy splat target. The PRODUCT transport is RCCL over xGMI (RcclTransport in testbed_main.cpp, three
// communicators so that the three exchanges of a step may run beside each other). StagedTransport below is TEST INFRASTRUCTURE for the same call sequence:
// every collective is staged through host memory and a directory shared by the ranks (one file per rank and collective). It exists so that the multi-rank
// code path -- non-zero chunk offsets, three blocks, sync_parameters() -- can run where RCCL cannot: two ranks sharing ONE GPU (RCCL refuses that), or the
// CPU-checker build of the same source (no device at all). It is selected by RNB_DP_TRANSPORT=staged + RNB_DP_STAGE_DIR and never by default.
#pragma once
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace dist {

enum DType { F32 = 0, F16 = 1, F64 = 2, U32 = 3, I32 = 4 };
enum Op { SUM = 0, MAX = 1 };
inline size_t dtype_size(DType t) { return t == F16 ? 2 : t == F64 ? 8 : 4; }

// `chan` 0 / 1 / 2: main, early block, step vector -- independent orders (RCCL orders the operations of ONE communicator whatever streams they are given).
// Every call is ordered on `stream` (a hipStream_t as void*; null in the CPU-checker build) like a kernel launch would be.
struct Transport {
	virtual ~Transport() {}
	virtual const char* name() const = 0;
	virtual int n_ranks() const = 0;
	virtual void all_reduce(void* buf, size_t n, DType t, Op op, int chan, void* stream) = 0;
	// block = world x chunk elements; own = block + rank x chunk receives the reduced chunk (in place)
	virtual void reduce_scatter(void* block, void* own, size_t chunk, DType t, int chan, void* stream) = 0;
	virtual void all_gather(const void* own, void* block, size_t chunk, DType t, int chan, void* stream) = 0;
	virtual void shutdown() {}
};

// How the staged transport reaches a buffer: device memory behind a stream (HIP build) or plain host memory (CPU-checker build).
struct MemOps {
	void (*to_host)(void* dst_host, const void* src, size_t bytes, void* stream);   // waits for `stream` first: the data is final
	void (*from_host)(void* dst, const void* src_host, size_t bytes, void* stream); // complete when it returns
};

namespace detail {
inline float h2f(uint16_t h) {
	const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu;
	uint32_t man = h & 0x3ffu, bits;
	if (exp == 0) {
		if (man == 0) bits = sign;
		else { int e = -1; do { ++e; man <<= 1; } while ((man & 0x400u) == 0); bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13; }
	} else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
	else bits = sign | (exp + 127 - 15) << 23 | man << 13;
	float f; std::memcpy(&f, &bits, 4); return f;
}
inline uint16_t f2h(float f) { // round to nearest even
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
// acc (op)= x, element-wise; ranks are folded in rank order on every rank, so all ranks hold the same bits (RCCL's ring makes no such promise; two ranks: a + b = b + a)
inline void fold(void* acc, const void* x, size_t n, DType t, Op op) {
	switch (t) {
		case F32: { float* a = (float*)acc; const float* b = (const float*)x; for (size_t i = 0; i < n; ++i) a[i] = op == SUM ? a[i] + b[i] : (a[i] < b[i] ? b[i] : a[i]); break; }
		case F64: { double* a = (double*)acc; const double* b = (const double*)x; for (size_t i = 0; i < n; ++i) a[i] = op == SUM ? a[i] + b[i] : (a[i] < b[i] ? b[i] : a[i]); break; }
		case F16: { uint16_t* a = (uint16_t*)acc; const uint16_t* b = (const uint16_t*)x; for (size_t i = 0; i < n; ++i) { const float u = h2f(a[i]), v = h2f(b[i]); a[i] = f2h(op == SUM ? u + v : (u < v ? v : u)); } break; }
		case U32: { uint32_t* a = (uint32_t*)acc; const uint32_t* b = (const uint32_t*)x; for (size_t i = 0; i < n; ++i) a[i] = op == SUM ? a[i] + b[i] : (a[i] < b[i] ? b[i] : a[i]); break; }
		case I32: { int32_t* a = (int32_t*)acc; const int32_t* b = (const int32_t*)x; for (size_t i = 0; i < n; ++i) a[i] = op == SUM ? a[i] + b[i] : (a[i] < b[i] ? b[i] : a[i]); break; }
	}
}
} // namespace detail

class StagedTransport : public Transport {
	std::string dir_;
	int world_, rank_;
	MemOps mem_;
	uint64_t seq_[3] = {0, 0, 0};
	double timeout_s_;
	std::vector<char> mine_, other_;

	std::string path(int chan, uint64_t seq, int rank) const { return dir_ + "/c" + std::to_string(chan) + "." + std::to_string(seq) + "." + std::to_string(rank); }
	void publish(int chan, uint64_t seq, const void* data, size_t bytes) {
		const std::string p = path(chan, seq, rank_), tmp = p + ".tmp";
		std::FILE* f = std::fopen(tmp.c_str(), "wb");
		if (!f || (bytes && std::fwrite(data, 1, bytes, f) != bytes)) { if (f) std::fclose(f); throw std::runtime_error("staged transport: cannot write " + tmp); }
		std::fclose(f);
		if (std::rename(tmp.c_str(), p.c_str()) != 0) throw std::runtime_error("staged transport: cannot publish " + p);
	}
	void fetch(int chan, uint64_t seq, int rank, void* data, size_t bytes) {
		const std::string p = path(chan, seq, rank);
		const auto t0 = std::chrono::steady_clock::now();
		for (;;) {
			struct stat st;
			if (::stat(p.c_str(), &st) == 0 && (size_t)st.st_size == bytes) {
				std::FILE* f = std::fopen(p.c_str(), "rb");
				if (f) {
					const bool ok = bytes == 0 || std::fread(data, 1, bytes, f) == bytes;
					std::fclose(f);
					if (ok) return;
				}
			}
			if (::access((dir_ + "/abort").c_str(), F_OK) == 0) throw std::runtime_error("staged transport: another rank has aborted the job");
			if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s_) throw std::runtime_error("staged transport: rank " + std::to_string(rank) + " did not arrive at " + p);
			::usleep(200);
		}
	}
	// every rank has published collective `seq` of this channel => every rank has finished reading collective seq - 1: this rank's file of it can go
	void retire(int chan, uint64_t seq) { if (seq >= 1) ::unlink(path(chan, seq - 1, rank_).c_str()); }

public:
	StagedTransport(const std::string& dir, int world, int rank, MemOps mem, double timeout_s = 120.0) : dir_(dir), world_(world), rank_(rank), mem_(mem), timeout_s_(timeout_s) {
		if (dir.empty()) throw std::runtime_error("staged transport: RNB_DP_STAGE_DIR is not set");
		::mkdir(dir.c_str(), 0777);
	}
	const char* name() const override { return "staged (host files; test transport)"; }
	int n_ranks() const override { return world_; }
	// a rank that fails tells the others, which would otherwise wait for its next file until the timeout
	void abort_job() { if (std::FILE* f = std::fopen((dir_ + "/abort").c_str(), "wb")) std::fclose(f); }

	void all_reduce(void* buf, size_t n, DType t, Op op, int chan, void* stream) override {
		const size_t bytes = n * dtype_size(t);
		const uint64_t seq = seq_[chan]++;
		mine_.resize(bytes); other_.resize(bytes);
		mem_.to_host(mine_.data(), buf, bytes, stream);
		publish(chan, seq, mine_.data(), bytes);
		std::vector<char> acc(bytes);
		for (int r = 0; r < world_; ++r) {
			if (r == rank_) { if (r == 0) std::memcpy(acc.data(), mine_.data(), bytes); else detail::fold(acc.data(), mine_.data(), n, t, op); continue; }
			fetch(chan, seq, r, other_.data(), bytes);
			if (r == 0) std::memcpy(acc.data(), other_.data(), bytes); else detail::fold(acc.data(), other_.data(), n, t, op);
		}
		retire(chan, seq);
		mem_.from_host(buf, acc.data(), bytes, stream);
	}
	void reduce_scatter(void* block, void* own, size_t chunk, DType t, int chan, void* stream) override {
		// (staged: every rank publishes its whole block and folds its own chunk of every rank's -- what a reduce-scatter leaves in `own`)
		const size_t cb = chunk * dtype_size(t), bytes = cb * (size_t)world_;
		const uint64_t seq = seq_[chan]++;
		mine_.resize(bytes); other_.resize(bytes);
		mem_.to_host(mine_.data(), block, bytes, stream);
		publish(chan, seq, mine_.data(), bytes);
		std::vector<char> acc(cb);
		for (int r = 0; r < world_; ++r) {
			const char* src = mine_.data();
			if (r != rank_) { fetch(chan, seq, r, other_.data(), bytes); src = other_.data(); }
			if (r == 0) std::memcpy(acc.data(), src + cb * rank_, cb); else detail::fold(acc.data(), src + cb * rank_, chunk, t, SUM);
		}
		retire(chan, seq);
		mem_.from_host(own, acc.data(), cb, stream);
	}
	void all_gather(const void* own, void* block, size_t chunk, DType t, int chan, void* stream) override {
		const size_t cb = chunk * dtype_size(t);
		const uint64_t seq = seq_[chan]++;
		mine_.resize(cb);
		mem_.to_host(mine_.data(), own, cb, stream);
		publish(chan, seq, mine_.data(), cb);
		std::vector<char> all(cb * (size_t)world_);
		for (int r = 0; r < world_; ++r) {
			if (r == rank_) std::memcpy(all.data() + cb * r, mine_.data(), cb);
			else fetch(chan, seq, r, all.data() + cb * r, cb);
		}
		retire(chan, seq);
		mem_.from_host(block, all.data(), all.size(), stream);
	}
	void shutdown() override {
		for (int c = 0; c < 3; ++c) if (seq_[c]) {
			// a last, empty collective per used channel: when it completes every rank has read everything, and the files left are this rank's last two
			const uint64_t seq = seq_[c]++;
			try { publish(c, seq, nullptr, 0); for (int r = 0; r < world_; ++r) if (r != rank_) fetch(c, seq, r, nullptr, 0); } catch (...) {}
			retire(c, seq);
		}
	}
};

} // namespace dist
