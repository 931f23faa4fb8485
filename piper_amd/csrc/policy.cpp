// The knob table of the launch policy (policy.h): one row per environment variable.
#include "policy.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>

namespace pe {

static const LaunchPolicy::Knob kKnobs[] = {
    {"PIPER_HIP_MRF", &LaunchPolicy::mrf, 0, 2, "fused MRF stage kernel: 0 off (conv by conv), 1 by the measured policy (ResBlock2 stages always; ResBlock1 stages on 32 channels up to PIPER_HIP_MRF_MAXF frames per call), 2 wherever it applies"},
    {"PIPER_HIP_MRF_MAXF", &LaunchPolicy::mrf_maxf, 0, 1L << 40, "batch frames up to which a 32-channel ResBlock1 stage runs in the fused kernel (mode 1)"},
    {"PIPER_HIP_MRF_OU", &LaunchPolicy::mrf_ou, 0, 4, "force the output units per wave of mrf_kernel (1..4) instead of the cost model; ignored when the width does not fit"},
    {"PIPER_HIP_MRF_TAIL", &LaunchPolicy::mrf_tail, 0, 1, "generator tail inside the last stage's mrf_kernel (0: separate conv_post_kernel)"},
    {"PIPER_HIP_MRF_SPLIT", &LaunchPolicy::mrf_split, 0, 1, "matrix modes bf16x3 / f16x3: the fused MRF stage kernel on the 16-bit matrix pipe with split operands (mrf_split_kernel) wherever a fused stage applies; 0 = the f32 fused kernel below PIPER_HIP_BF3_MINF frames, conv by conv on the 16-bit pipe above"},
    {"PIPER_HIP_BF3_MINF", &LaunchPolicy::bf3_minf, 0, 1L << 40, "split matrix modes without a fused 16-bit stage kernel (bf16x6, or PIPER_HIP_MRF_SPLIT=0): batch frames from which the <= 64-channel MRF stages run conv by conv on the 16-bit pipe"},
    {"PIPER_HIP_SPLITK_MAX", &LaunchPolicy::splitk_max, 0, 1L << 40, "tile-workgroup count below which a conv goes to the split-K kernels (0: always the tiled kernel)"},
    {"PIPER_HIP_SPLITK16", &LaunchPolicy::splitk16, 0, 3, "16-column split-K form: 0 off, 1 gate convs, 2 + long-K plain convs, 3 everywhere"},
    {"PIPER_HIP_WIDE_SPLITK", &LaunchPolicy::wide_splitk, 0, 2, "12-wave split-K: 0 off, 1 WN gate conv, 2 always"},
    {"PIPER_HIP_TPB", &LaunchPolicy::tpb, 0, 64, "tiled kernel: column tiles walked by one workgroup (0 = 1)"},
    {"PIPER_HIP_GROUP_MRF", &LaunchPolicy::group_mrf, 0, 2, "sibling resblock convs of a wider small stage as grouped launches (one-utterance calls): 0 off, 2 without the K-concatenated last step"},
    {"PIPER_HIP_GROUP_MAXB", &LaunchPolicy::group_maxb, 0, 1L << 40, "64 x 64 tiles of one conv of a stage below which the grouped sibling launches of a single utterance take the split-K kernels (the tiled grouped launches from there on)"},
    {"PIPER_HIP_GROUP_TILED", &LaunchPolicy::group_tiled, 0, 1, "sibling resblock convs of a stage that runs the tiled kernel as grouped launches (conv_mfma_group_kernel) while one conv is at most 2048 tile workgroups (8 per CU): 0 = one launch per conv"},
    {"PIPER_HIP_COLCHAIN", &LaunchPolicy::colchain, 0, 2, "colchain_kernel / lngemm_kernel: 0 off, 1 up to 4096 ids / 8192 frames per call, 2 always"},
    {"PIPER_HIP_COL4", &LaunchPolicy::col4, 0, 2, "4-column forms of the 192-channel chains: 0 off, 1 up to PIPER_HIP_COL4_MAXC ids (2048 frames for the flow's launches) per call, 2 always"},
    {"PIPER_HIP_COL4_MAXC", &LaunchPolicy::col4_maxc, 0, 1L << 40, "ids per call up to which the 4-column chains are used"},
    {"PIPER_HIP_FFN", &LaunchPolicy::ffn, 0, 1, "the encoder FFN as one launch (ffn_kernel) wherever the 4-column chains run: 0 = conv by conv"},
    {"PIPER_HIP_ATTNO", &LaunchPolicy::attno, 0, 1, "attention + conv_o + norm_layers_1 as one launch (attno_kernel) wherever the 4-column chains run: 0 = attn_kernel + colchain4_kernel"},
    {"PIPER_HIP_ATTN4", &LaunchPolicy::attn4, 0, 2, "attention + conv_o + norm_layers_1 on 4-query workgroups (attn4_kernel) in place of attno_kernel: 0 off, 1 for calls whose longest utterance has up to PIPER_HIP_ATTN4_MAXC ids, 2 wherever attno_kernel applies"},
    {"PIPER_HIP_ATTN4_MAXC", &LaunchPolicy::attn4_maxc, 0, 1L << 40, "ids of the longest utterance of a call up to which attention runs on 4-query workgroups"},
    {"PIPER_HIP_FUSE_DP", &LaunchPolicy::fuse_dp, 0, 1, "ConvFlow.pre / proj / spline fused into the DDSConv layer launches"},
    {"PIPER_HIP_SPEC", &LaunchPolicy::spec, 0, 1, "speculative stage-B sizing / whole utterance as one graph for <= 4 utterances per call"},
    {"PIPER_HIP_SPEC_EXPECT", &LaunchPolicy::spec_expect, 0, 1, "speculative graphs planned for the expected frame counts (0: for the bucket capacity)"},
    {"PIPER_HIP_PCM_ZC", &LaunchPolicy::pcm_zc, 0, 1, "PCM written straight into pinned host memory by pcm16_kernel (0: one copy per utterance behind the graph)"},
    {"PIPER_HIP_IDS_ZC", &LaunchPolicy::ids_zc, 0, 1, "phoneme ids, lengths and speaker ids read straight from the pinned host block by embed_kernel, up to 65536 padded ids per call (0: one host-to-device copy in front of the graph)"},
    {"PIPER_HIP_NO_GRAPH", &LaunchPolicy::no_graph, 0, 1, "launch kernels directly instead of replaying hipGraphs"},
    {"PIPER_HIP_GRAPHS", &LaunchPolicy::graphs, 1, 4096, "hipGraphs kept per engine (least recently used evicted one at a time)"},
    {"PIPER_HIP_CONVT_VEC", &LaunchPolicy::convt_vec, 0, 1, "polyphase up-conv: a lane's four accumulator rows leave as one 16-byte store (stride a multiple of 4) or two 8-byte stores (stride 2) of consecutive output samples; 0: one 4-byte store per phase"},
    {"PIPER_HIP_XCD", &LaunchPolicy::xcd, -1, 32, "XCDs the dispatch round-robins over, for the XCD-aware tile orders (-1 = probed at engine creation, 0 = tiles in workgroup order)"},
    {"PIPER_HIP_XCD_FFN", &LaunchPolicy::xcd_ffn, 0, 1, "ffn_kernel deals (column tile, slice) to the XCDs slice-major, so an XCD's L2 holds two slices' weights instead of all sixteen (0: blockIdx order)"},
    {"PIPER_HIP_STACK_PRE", &LaunchPolicy::stack_pre, 0, 1, "small calls of the 192-channel voices: enc_p.proj and dp.pre (both read the last LayerNorm's output) as one lngemm4_kernel launch over the stacked matrix (0: two launches)"},
    {"PIPER_HIP_CHAIN_RS", &LaunchPolicy::chain_rs, 0, 1, "small calls of the 192-channel voices: the last WN layer's res/skip conv (skip rows only) in front of the coupling layer's post + pre chain launch, colchain4_kernel<true> (0: a launch of its own)"},
    {"PIPER_HIP_GATE4", &LaunchPolicy::gate4, 0, 2, "short calls: the WN gate conv over 192 channels on 64-row x 12-column workgroups with the 4x4x1 MFMA (gate4_kernel) up to 640 workgroups: 0 off (16-column split-K form), 2 wherever the split-K route applies"},
    {"PIPER_HIP_GATE_HALF", &LaunchPolicy::gate_half, 0, 1, "short one-utterance calls (up to 128 whole-group workgroups): the WN gate conv on half a 32-channel group per workgroup, six waves with the whole K range in flight, twice the workgroups (0: whole groups on twelve waves)"},
    {"PIPER_HIP_CONV1X1", &LaunchPolicy::conv1x1, 0, 1, "batched one-tap convs (q/k/v, WN res/skip, coupling pre/post, proj) through conv1x1_kernel, B operand straight from global memory (0: the tiled kernel)"},
    {"PIPER_HIP_WS_BUDGET_MB", &LaunchPolicy::ws_budget_mb, 0, 1048576, "MiB of device memory the workspace of one pipeline half may take (0: a third of the device's memory): capacities grow only inside it, a call that does not fit by itself is an error"},
    {"PIPER_HIP_ATTN_LONG", &LaunchPolicy::attn_long, 0, 1, "attention score slabs in global memory (attn_long_kernel) at every length, also in place of attno_kernel (tests): by default only utterances whose 32 x T slab does not fit LDS (more than ~830 ids) take that form"},
    {"PIPER_HIP_PROF_SITES", &LaunchPolicy::prof_sites, 0, 1, "level-2 profile rows of the tiled conv kernel per conv shape (tuning aid)"},
    {"PIPER_HIP_DEBUG_KEEP", &LaunchPolicy::debug_keep, 0, 1, "test hook: keep z_p for pe_debug_tensor"},
    {"PIPER_HIP_DEBUG_POISON", &LaunchPolicy::debug_poison, 0, 1, "test hook: activation workspaces are filled with NaN bit patterns when allocated (no kernel may read what the call did not write)"},
};

const LaunchPolicy::Knob* LaunchPolicy::knobs(int* n) {
  if (n) *n = (int)(sizeof(kKnobs) / sizeof(kKnobs[0]));
  return kKnobs;
}

void LaunchPolicy::read_env() {
  for (const Knob& k : kKnobs) {
    const char* t = getenv(k.env);
    if (!t || !*t) continue;
    errno = 0;
    char* end = nullptr;
    const long v = strtol(t, &end, 10);
    if (errno || end == t || *end)
      throw std::runtime_error(std::string(k.env) + "=" + t + ": not an integer (" + k.doc + ")");
    this->*k.field = v < k.lo ? k.lo : (v > k.hi ? k.hi : v);
  }
}

int LaunchPolicy::matrix_mode_env() {
  const char* t = getenv("PIPER_HIP_MATRIX");
  if (!t || !t[0] || !strcmp(t, "f32")) return -1;
  if (!strcmp(t, "bf16x3")) return 0;
  if (!strcmp(t, "f16x3")) return 1;
  if (!strcmp(t, "bf16x6")) return 2;
  throw std::runtime_error("PIPER_HIP_MATRIX: expected f32, bf16x3, f16x3 or bf16x6");
}

const char* LaunchPolicy::describe() {
  static std::string out;
  static std::once_flag once;
  std::call_once(once, [] {
    const LaunchPolicy d;
    out = "[";
    for (const Knob& k : kKnobs) {
      char head[160];
      snprintf(head, sizeof(head), "%s{\"env\":\"%s\",\"default\":%ld,\"lo\":%ld,\"hi\":%ld,\"doc\":\"", out.size() > 1 ? "," : "", k.env,
               d.*k.field, k.lo, k.hi);
      out += head;
      for (const char* c = k.doc; *c; ++c) {
        if (*c == '"' || *c == '\\') out += '\\';
        out += *c;
      }
      out += "\"}";
    }
    out += "]";
  });
  return out.c_str();
}

}  // namespace pe
