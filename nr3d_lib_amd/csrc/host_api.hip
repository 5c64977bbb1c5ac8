// nr3d_lib_amd/csrc/host_api.hip -- host-only parts of the C ABI: error string, ABI version and the
// LoTD meta builder (reference: LoDMeta::create_meta, csrc/lotd/src/lotd_torch_api.cu:29-230).
#include "common.h"
#include <stdlib.h>
#include <utility>
#include <string.h>
#include <limits>
#include <chrono>
#include <mutex>
#include <vector>

namespace nr3d {

static thread_local char g_err[512] = {0};

char *err_buf() { return g_err; }

int fail(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return 1;
}

namespace prof {
uint32_t g_mask = 0;
struct Pair { hipEvent_t a, b; };
static std::mutex g_mu;
static std::vector<Pair> g_done[NR3D_PROF_COUNT];     // recorded, not yet read
static std::vector<Pair> g_free;                      // event pairs for reuse
static Pair g_open[NR3D_PROF_COUNT];
static bool g_is_open[NR3D_PROF_COUNT] = {};

void begin(int id, hipStream_t st) {
	std::lock_guard<std::mutex> lk(g_mu);
	if (id < 0 || id >= NR3D_PROF_COUNT || g_is_open[id]) return;
	Pair p;
	if (!g_free.empty()) { p = g_free.back(); g_free.pop_back(); }
	else if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
	if (hipEventRecord(p.a, st) != hipSuccess) return;
	g_open[id] = p; g_is_open[id] = true;
}
void end(int id, hipStream_t st) {
	std::lock_guard<std::mutex> lk(g_mu);
	if (id < 0 || id >= NR3D_PROF_COUNT || !g_is_open[id]) return;
	g_is_open[id] = false;
	if (hipEventRecord(g_open[id].b, st) == hipSuccess) g_done[id].push_back(g_open[id]);
}
}  // namespace prof

namespace opt {
static const int64_t kDefault[NR3D_OPT_COUNT] = {
	/* LOTD_PAIR */ 1, /* PAIR_QUAD */ 1, /* PAIR_SECOND */ 1, /* PAIR_DIRECT */ 1, /* PAIR_FIXED */ 1, /* FWD_PAIRLANE */ 1,
	/* FWD_SPLIT */ 1, /* FWD_LDS_STAGE */ 1, /* HVP_LEVELS */ 1, /* HVP_PAIRLANE */ 1, /* HVP_SPLIT */ 1, /* VM_SPLIT */ 1,
	/* CP_DIRECT */ 1, /* MARCH_GROUP */ 0, /* PACK_SCAN */ 1, /* VM_LINES_DIRECT */ 0, /* FWD_CELL_MAJOR */ 1, /* SORT_WAVE */ 1,
	/* VM_DIRECT */ 1, /* DIRECT_FIXED */ 1, /* VM_SORTED */ 1, /* MLP_X3 */ 1,
};
std::atomic<int64_t> g_val[NR3D_OPT_COUNT] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 0, 1, 1, 1, 1, 1, 1};

#ifdef NR3D_EXPERIMENTS
// measurement knobs of the experiments build: NR3D_<NAME> from the environment, looked up once per name
int64_t experiment_env(const char *name, int64_t dflt) {
	static std::mutex mu;
	static std::vector<std::pair<const char *, int64_t>> seen;
	std::lock_guard<std::mutex> lk(mu);
	for (const auto &kv : seen) if (kv.first == name) return kv.second;        // string literals: pointer identity
	const char *e = getenv(name);
	const int64_t v = e ? (int64_t)atoll(e) : dflt;
	seen.emplace_back(name, v);
	return v;
}
#endif
}  // namespace opt

}  // namespace nr3d

using namespace nr3d;

extern "C" void nr3d_prof_enable(uint32_t mask) { prof::g_mask = mask; }

extern "C" int nr3d_prof_read(int id, double *total_ms, uint32_t *n_intervals, int reset) {
	NR3D_CHECK(id >= 0 && id < NR3D_PROF_COUNT && total_ms && n_intervals, "prof_read: bad argument");
	std::lock_guard<std::mutex> lk(prof::g_mu);
	double sum = 0.0;
	for (const prof::Pair &p : prof::g_done[id]) {
		NR3D_HIP_CHECK(hipEventSynchronize(p.b));
		float ms = 0.0f;
		NR3D_HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
		sum += ms;
	}
	*total_ms = sum;
	*n_intervals = (uint32_t)prof::g_done[id].size();
	if (reset) {
		for (const prof::Pair &p : prof::g_done[id]) prof::g_free.push_back(p);
		prof::g_done[id].clear();
	}
	return 0;
}

extern "C" int nr3d_set_option(int id, int64_t value) {
	NR3D_CHECK(id >= 0 && id < NR3D_OPT_COUNT, "nr3d_set_option: unknown option %d", id);
	opt::g_val[id].store(value < 0 ? opt::kDefault[id] : value, std::memory_order_relaxed);
	return 0;
}
extern "C" int64_t nr3d_get_option(int id) { return (id >= 0 && id < NR3D_OPT_COUNT) ? opt::get(id) : -1; }

extern "C" int nr3d_wait_host_words(const int64_t *words, int n, int64_t sentinel, uint32_t timeout_us) {
	if (!words || n <= 0) return 0;
	const auto t0 = std::chrono::steady_clock::now();
	for (uint32_t spin = 0;; ++spin) {
		bool all = true;
		for (int i = 0; i < n; ++i)
			if (__atomic_load_n(&words[i], __ATOMIC_ACQUIRE) == sentinel) { all = false; break; }
		if (all) return 0;
		__builtin_ia32_pause();
		if ((spin & 63u) == 63u &&
		    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > (int64_t)timeout_us)
			return 1;
	}
}

extern "C" const char *nr3d_last_error(void) { return err_buf(); }
extern "C" int nr3d_abi_version(void) { return NR3D_ABI_VERSION; }

extern "C" uint64_t nr3d_sort_pairs_u32_tmp_bytes(uint32_t n_max, int batch) { return (uint64_t)rsort::tmp_bytes(n_max, batch == 2 ? 2 : 1); }
extern "C" int nr3d_sort_pairs_u32(void *tmp, int batch, const uint32_t *kin0, const uint32_t *vin0, uint32_t *kout0, uint32_t *vout0,
                                   const uint32_t *kin1, const uint32_t *vin1, uint32_t *kout1, uint32_t *vout1, uint32_t n_max,
                                   const uint32_t *n_dev, int bits, void *stream) {
	NR3D_CHECK(batch == 1 || batch == 2, "nr3d_sort_pairs_u32: batch must be 1 or 2");
	NR3D_CHECK(n_max == 0 || (tmp && kin0 && kout0 && vout0 && (batch == 1 || (kin1 && kout1 && vout1))), "nr3d_sort_pairs_u32: NULL argument");
	const uint32_t *kin[2] = {kin0, kin1}, *vin[2] = {vin0, vin1};
	uint32_t *kout[2] = {kout0, kout1}, *vout[2] = {vout0, vout1};
	return rsort::sort_pairs(tmp, batch, kin, vin, kout, vout, n_max, n_dev, bits, (hipStream_t)stream);
}

extern "C" int nr3d_lotd_meta_create(int32_t n_input_dim, uint32_t n_levels, const int32_t *res_multidim,
                                     const int32_t *n_feats, const int32_t *types, uint32_t hashmap_size,
                                     int use_smooth_step, nr3d_lotd_meta_t *m) {
	NR3D_CHECK(m && res_multidim && n_feats && types, "LoTDEncoding: NULL argument");
	memset(m, 0, sizeof(*m));
	NR3D_CHECK(n_input_dim == 2 || n_input_dim == 3 || n_input_dim == 4, "LoTDEncoding: `n_input_dim` must be 2/3/4.");
	NR3D_CHECK(n_levels <= NR3D_LOTD_MAX_LEVELS, "LoTDEncoding:` num_level`=%u exceeds maximum level=%d", n_levels,
	           NR3D_LOTD_MAX_LEVELS);
	const uint32_t D = (uint32_t)n_input_dim;
	m->n_dims_to_encode = D;
	m->n_levels = n_levels;
	m->interpolation_type = use_smooth_step ? 1u : 0u;

	// feature width of a pseudo level = largest of {8,4,2} dividing every level's width
	auto divides_all = [&](int32_t k) {
		for (uint32_t l = 0; l < n_levels; ++l) if (n_feats[l] % k != 0) return false;
		return true;
	};
	uint32_t G = 0;
	for (int32_t k : {8, 4, 2}) if (divides_all(k)) { G = (uint32_t)k; break; }
	NR3D_CHECK(G != 0, "LoTDEncoding: the greatest common divisor of `lod_n_feats` must be at least 2");
	m->n_feat_per_pseudo_lvl = G;

	const float max_params = (float)(std::numeric_limits<uint32_t>::max() / 2);
	uint32_t total = 0, n_pseudo = 0, n_enc = 0;
	float total_f = 0.0f;
	bool dense_hash_only = true;
	for (uint32_t l = 0; l < n_levels; ++l) {
		nr3d_lotd_level_t &L = m->levels[l];
		const uint32_t tp = (uint32_t)types[l];
		NR3D_CHECK(tp <= NR3D_LOD_Hash, "LoTDEncoding: Invalid lod type: %u", tp);
		NR3D_CHECK(n_feats[l] > 0, "LoTDEncoding: n_feats must be positive");
		if (tp != NR3D_LOD_Dense && tp != NR3D_LOD_Hash) dense_hash_only = false;
		uint64_t vol = 1, sum_r = 0;
		for (uint32_t d = 0; d < D; ++d) {
			const int32_t r = res_multidim[l * D + d];
			NR3D_CHECK(r > 2, "LoTDEncoding: only support grid resolutions >= 3");
			L.res[d] = (uint32_t)r;
			vol *= (uint64_t)r;
			sum_r += (uint64_t)r;
		}
		// entries per level; plane_d = volume / R_d
		uint64_t sum_planes = 0;
		for (uint32_t d = 0; d < D; ++d) sum_planes += vol / L.res[d];
		uint64_t size = 0;
		switch (tp) {
		case NR3D_LOD_Dense: size = vol; break;
		case NR3D_LOD_NPlaneMul:
		case NR3D_LOD_NPlaneSum: size = sum_planes; break;
		case NR3D_LOD_VectorMatrix:
			NR3D_CHECK(D == 3, "LoTDEncoding: VectorMatrix mode only support 3D encoding.");
			size = sum_planes + sum_r;
			break;
		case NR3D_LOD_VecZMatXoY:
			NR3D_CHECK(D == 3, "LoTDEncoding: VecZMatXoY mode only support 3D encoding.");
			size = (uint64_t)L.res[0] * L.res[1] + L.res[2];
			break;
		case NR3D_LOD_CP:
		case NR3D_LOD_CPfast: size = sum_r; break;
		case NR3D_LOD_Hash:
			NR3D_CHECK(hashmap_size != 0, "LoTDEncoding: Hash mode need `hashmap_size`");
			size = hashmap_size;
			break;
		}
		total_f += (float)size * (float)n_feats[l];
		NR3D_CHECK(total_f <= max_params, "LoTDEncoding: param size too large.");
		L.n_feats = (uint32_t)n_feats[l];
		L.type = tp;
		L.size = (uint32_t)size;
		L.offset = total;
		total += (uint32_t)size * (uint32_t)n_feats[l];
		for (uint32_t j = 0; j < (uint32_t)n_feats[l] / G; ++j) {
			NR3D_CHECK(n_pseudo < NR3D_LOTD_MAX_PSEUDO, "LoTDEncoding: too many pseudo levels");
			m->map_levels[n_pseudo] = (uint16_t)l;
			m->map_cnt[n_pseudo] = (uint16_t)j;
			m->map_col[n_pseudo] = (uint16_t)(n_enc + j * G);
			++n_pseudo;
		}
		n_enc += (uint32_t)n_feats[l];
	}
	NR3D_CHECK(n_enc <= 1024, "LoTDEncoding: total number of features too large. Shoule be <= 1024.");
	m->n_params = total;
	m->n_pseudo_levels = n_pseudo;
	m->n_encoded_dims = n_enc;
	m->c_hash_only = dense_hash_only ? 1u : 0u;
	return 0;
}

extern "C" int nr3d_lotd_meta_regroup(const nr3d_lotd_meta_t *meta, uint32_t width, uint32_t max_width, nr3d_lotd_meta_t *out) {
	NR3D_CHECK(meta && out, "LoTD::meta_regroup: NULL argument");
	NR3D_CHECK(width == 2 || width == 4 || width == 8, "LoTD::meta_regroup: width must be 2, 4 or 8");
	*out = *meta;
	out->n_feat_per_pseudo_lvl = width;
	memset(out->map_levels, 0, sizeof(out->map_levels));
	memset(out->map_cnt, 0, sizeof(out->map_cnt));
	memset(out->map_col, 0, sizeof(out->map_col));
	uint32_t n_pseudo = 0, col = 0;
	for (uint32_t l = 0; l < meta->n_levels; ++l) {
		const uint32_t F = meta->levels[l].n_feats;
		uint32_t best = (F % 8u == 0u) ? 8u : (F % 4u == 0u) ? 4u : 2u;
		if (max_width >= 2u && best > max_width) best = max_width;
		if (best == width)
			for (uint32_t j = 0; j < F / width; ++j) {
				NR3D_CHECK(n_pseudo < NR3D_LOTD_MAX_PSEUDO, "LoTD::meta_regroup: too many pseudo levels");
				out->map_levels[n_pseudo] = (uint16_t)l;
				out->map_cnt[n_pseudo] = (uint16_t)j;
				out->map_col[n_pseudo] = (uint16_t)(col + j * width);
				++n_pseudo;
			}
		col += F;
	}
	out->n_pseudo_levels = n_pseudo;
	return 0;
}
