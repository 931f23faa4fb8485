/* piper_hip.h -- C ABI of the MI355X-native VITS synthesis engine (libpiper_hip.so).
 *
 * This is the drop-in boundary for the ONNX Runtime session that the reference's
 * piper::synthesize() drives (reference src/cpp/piper.cpp:337-441). Each entry point names the
 * reference interface it replaces. Plain pointers and sizes only; all functions return 0 on success
 * and a non-zero code on failure, with the message available from pe_last_error() (the reference
 * throws Ort::Exception / std::runtime_error at the same places; the C++ shim in
 * piper_amd/csrc/piper.hpp re-throws std::runtime_error).
 *
 * Threading: like the reference an engine handle is not thread-safe; use one
 * handle per thread / per GPU. Output buffers returned by pe_synthesize* are owned by the engine and
 * stay valid until the next call on the same handle.
 */
#ifndef PIPER_HIP_H_
#define PIPER_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pe_engine pe_engine;

/* Optional injected N(0,1) draws for the two sampling sites of the graph (reference
 * vits/models.py:111 and :718; ONNX RandomNormalLike nodes). NULL pointers -> the engine's own
 * counter-based generator (seed: pe_set_seed). Host pointers.
 *   noise_w: [B][2][w_stride]        column t is used for phoneme id t
 *   noise_z: [B][inter][z_stride]    column f is used for frame f (z_stride >= frames) */
typedef struct pe_noise {
  const float* noise_w;
  int64_t w_stride;
  const float* noise_z;
  int64_t z_stride;
} pe_noise;

/* Result views of the last synthesis call (engine-owned host memory). */
typedef struct pe_result {
  int32_t batch;
  const int64_t* sample_offsets; /* [batch+1] prefix offsets into audio / pcm */
  const float* audio;            /* float waveform in [-1,1], what Ort "output" [1,1,1,S] held (piper.cpp:397-400) */
  const int16_t* pcm;            /* peak-normalised int16 as piper.cpp:410-431 / util.py:5-12 produce */
  const int32_t* frames;         /* [batch] spectrogram frames per utterance (S = frames * hop) */
  double infer_seconds;          /* wall time of the device pipeline, the reference's inferSeconds (piper.cpp:385-395) */
} pe_result;

/* Replaces Ort::Env + SessionOptions + Ort::Session(model path) in loadModel()
 * (piper.cpp:262-306): parses the voice .onnx (export_onnx.py graph), packs the weights for the
 * MFMA kernels and uploads them to GPU `device`. `onnx_path` may also name the output of the reference's
 * streaming export (export_onnx_streaming.py: a directory with encoder.onnx + decoder.onnx, or either of
 * the two files): both graphs are read and the voice behaves like the single-file one. */
int pe_create(const char* onnx_path, int device, pe_engine** out);

/* Same from an in-memory weight blob (PEBLOB01, see piper_amd/weights.py). This is also what the
 * other ranks of a multi-GPU job call after the RCCL broadcast of rank 0's blob. */
int pe_create_from_blob(const void* blob, size_t nbytes, int device, pe_engine** out);

/* Multi-GPU loading without a host round trip per rank (one RCCL broadcast of the voice at load).
 * The packed weights of an engine live in ONE device arena whose layout follows from the tensor SHAPES alone:
 *   every rank:  pe_weights_bound(header)            -> arena size to allocate (e.g. a torch / RCCL-registered buffer)
 *   root rank:   pe_create_in_arena(blob, .., arena, bytes, skeleton = 0)   parses, packs, uploads into its arena
 *   other ranks: pe_create_in_arena(header, .., arena, bytes, skeleton = 1) lays the arena out, touches no weight data
 *   all:         broadcast the first pe_weights_used() bytes of the root's arena into the others' arenas (ncclBroadcast)
 *   other ranks: pe_arena_ready()
 * `header` = the first 8 + 4*64 + 8 + n_tensors*136 bytes of a PEBLOB01 (magic, architecture, tensor records).
 * The arena must be 256-byte aligned and outlive the engine. */
int pe_weights_bound(const void* blob_or_header, size_t nbytes, size_t* bound);
int pe_create_in_arena(const void* blob_or_header, size_t nbytes, int device, void* arena, size_t arena_bytes,
                       int skeleton, pe_engine** out);
int pe_weights_used(pe_engine* e, size_t* used);
int pe_arena_ready(pe_engine* e);

/* .onnx -> PEBLOB01 in host memory (no GPU needed). Free with pe_free(). */
int pe_onnx_to_blob(const char* onnx_path, void** blob, size_t* nbytes);
void pe_free(void* p);

/* Replaces session.onnx.Run() for one utterance (piper.cpp:386-388) with the inputs of
 * piper.cpp:342-377: ids = "input"[1,T] (int64), scales = {noise_scale, length_scale, noise_w},
 * sid = "sid" or -1 for single-speaker voices. */
int pe_synthesize(pe_engine* e, const int64_t* ids, int64_t n_ids, const float scales[3], int64_t sid,
                  const pe_noise* noise, pe_result* result);

/* Batched form: B independent utterances, each computed exactly as a B=1 Run() would (no cross-talk
 * through padding). ids are concatenated; offsets[B+1] delimit them; sids may be NULL. */
int pe_synthesize_batch(pe_engine* e, const int64_t* ids, const int64_t* offsets, int32_t batch,
                        const float scales[3], const int64_t* sids, const pe_noise* noise, pe_result* result);

/* The same call in three parts (pe_synthesize_batch = pe_upload + pe_run + pe_fetch(1, 1)): upload = validate and stage the
 * inputs (ids / lengths / speaker ids of calls up to 65 536 padded ids stay in the engine's pinned host block and are read
 * in place by the first kernel -- no copy is enqueued; larger calls and injected noise are copied to HBM), run = the device
 * pipeline (may be repeated on the uploaded inputs: every run draws fresh noise), fetch = wait + result views (the int16 PCM
 * is written straight into pinned host memory by the last kernel; want_audio adds a copy of the float waveform). */
int pe_upload(pe_engine* e, const int64_t* ids, const int64_t* offsets, int32_t batch, const float scales[3],
              const int64_t* sids, const pe_noise* noise);
int pe_run(pe_engine* e);
int pe_fetch(pe_engine* e, int want_audio, int want_pcm, pe_result* result);

/* Streaming decode of one utterance (BASELINE.json configs[4]; reference behaviour:
 * src/python/piper_train/infer_onnx_streaming.py:76-124 -- encoder once, then the decoder on chunks of
 * frames). pe_stream_begin runs the text encoder, duration predictor and flow and reports the frame count;
 * every pe_stream_next returns the samples of the next `chunk_frames` frames (reference default 45),
 * decoded on a window padded by the generator's exact receptive half-width (`halo_frames`), so the chunks
 * concatenate to exactly the unchunked waveform. `pcm` is peak-normalised per chunk, like the reference's
 * streaming script. *n_samples == 0 means the utterance is finished. */
int pe_stream_begin(pe_engine* e, const int64_t* ids, int64_t n_ids, const float scales[3], int64_t sid,
                    const pe_noise* noise, int32_t* total_frames, int32_t* halo_frames);
int pe_stream_next(pe_engine* e, int32_t chunk_frames, const float** audio, const int16_t** pcm, int64_t* n_samples);

/* Integer per-id durations (ceil(w), reference models.py:703) of the last call, concatenated like ids. */
int pe_get_durations(pe_engine* e, int32_t* out, int64_t capacity, int64_t* n);

/* Voice facts the caller needs (sample rate from the architecture header of a blob, hop size, ...). */
int pe_get_info(pe_engine* e, int32_t* sample_rate, int32_t* hop, int32_t* n_speakers, int32_t* n_symbols,
                int64_t* weight_bytes);

void pe_set_seed(pe_engine* e, uint64_t seed);

/* Timing with HIP events on the engine's stream. level 1: one pair per pipeline stage (rows
 * text_encoder, duration_predictor, regulate+flow, hifigan, post+pcm). level 2: additionally one pair
 * around every conv / attention / layer-norm launch (rows named after the kernel), with the launch's
 * algorithmic FLOPs. ms/flops/launches accumulate until pe_profile_reset(); 0 switches it off. */
int pe_profile_enable(pe_engine* e, int level);
int pe_profile_reset(pe_engine* e);
int pe_profile_rows(pe_engine* e);
int pe_profile_get(pe_engine* e, int row, const char** name, double* ms, double* flops, int64_t* launches);
/* level 2 rows of the conv kernels also carry the launches' algorithmic HBM bytes (inputs + outputs + residual
 * operands + weights once), the denominator for comparing PMC-measured traffic against */
int pe_profile_bytes(pe_engine* e, int row, double* bytes);

/* The HIP stream (hipStream_t) the engine launches on, for callers that bracket it with their own events. */
void* pe_stream(pe_engine* e);

/* Test hook: copy an internal per-stage tensor of utterance b: x_enc, stats (rows m_p then logs_p, models.py:208),
 * xg, logw, z_p (only with PIPER_HIP_DEBUG_KEEP=1 in the environment at pe_create), z, noise_w, noise_z, audio. */
int pe_debug_tensor(pe_engine* e, const char* name, int32_t b, float* out, int64_t capacity, int32_t* rows,
                    int32_t* cols);

/* Test hooks for the N(0,1) generator behind the graph's two RandomNormalLike sites (models.py:111, :718) when no
 * noise is injected. A site's stream is a logical [row][65536] array, row = utterance * channels + channel (2 channels
 * at site 0, inter_channels at site 1), column = phoneme id / frame: the value the pipeline uses there depends on
 * (seed, run counter, site, row, column) only -- not on batch buckets or workspace sizes. pe_debug_randn fills out[n]
 * with draws row * 65536 .. + n of site 0/1 at run counter `call` under the current seed; pe_rng_calls is the number
 * of pipeline runs so far (the counter the next run will use is that + 1). */
int pe_debug_randn(pe_engine* e, int32_t site, uint64_t call, int64_t row, int64_t n, float* out);
uint64_t pe_rng_calls(pe_engine* e);

/* Kernel launches (hipGraph kernel nodes) the last pe_run / pe_synthesize* issued: the length of the dependent
 * launch chain one utterance costs (the latency figure of merit at batch 1). */
int64_t pe_run_launches(pe_engine* e);

/* Calls of <= 4 utterances enqueue the second half of the pipeline (flow + vocoder) for a GUESSED frame bucket right
 * behind the first half -- the frame count is the graph's only data-dependent shape (reference models.py:702-716) and
 * would otherwise cost a host round trip mid-pipeline. The guess (running maximum of frames per id x an adaptive
 * margin) is verified when the results are fetched; a miss re-runs the second half. runs = calls issued that way since
 * pe_create, misses = guesses that were too small (each cost one extra pass of the second half). */
int pe_speculation_stats(pe_engine* e, int64_t* runs, int64_t* misses);

/* Session warm-up -- what loadModel's session creation does for ORT (piper.cpp:262-306: graph optimisation at load), here
 * for the hipGraphs: the kernel sequence of a call is captured once per shape bucket (ids in steps of 32 up to 512, then
 * 8 steps per octave; frames in steps of 64 up to 1024, then 16 per octave) and replayed afterwards; the cache keeps the
 * 256 most recently used graphs (PIPER_HIP_GRAPHS) and evicts one at a time. pe_warmup
 *   - sizes the workspaces for calls of up to max_batch utterances x max_ids ids and frames_per_id * max_ids frames
 *     (<= 0: 8), so that no later call grows them (growth re-creates every graph), and
 *   - if sample_ids is given (a representative utterance of the voice: its frames-per-id ratio seeds the speculative
 *     sizing), synthesises it cut / tiled to every id bucket up to max_ids with `scales` (NULL: 0.667 / 1 / 0.8), so that
 *     the single-utterance graphs exist before the first real call.
 * pe_graph_stats: graphs currently cached / captures since pe_create (a steady server stops capturing). */
int pe_warmup(pe_engine* e, int32_t max_batch, int32_t max_ids, float frames_per_id, const float scales[3],
              const int64_t* sample_ids, int64_t n_sample);
int pe_graph_stats(pe_engine* e, int64_t* cached, int64_t* captures);

/* Diagnostic: which XCD (accelerator complex of the MI355X) ran workgroups 0..63 of a 1-D probe launch at pe_create
 * (xcc[64]), and *period = P when that was a round-robin over P XCDs (0 otherwise). The small-call kernels order their
 * column tiles by it (piper_amd/csrc/kernels/col4.h); bench.py prints it so that a result line says what the box did. */
int pe_xcc_pattern(pe_engine* e, int32_t xcc[64], int32_t* period);

/* Diagnostic: the PCI bus id ("0000:05:00.0") of HIP device `device` as this process sees it -- what a multi-GPU record
 * lists per rank so that a reader can check that N ranks ran on N DISTINCT devices (bench.py --gpus N). */
int pe_device_pci_bus_id(int device, char* out, int32_t capacity);

/* Diagnostic: the engine's launch-policy knobs -- every environment variable that picks a kernel form, with its default,
 * range and meaning -- as a JSON array of {"env", "default", "lo", "hi", "doc"} objects (piper_amd/csrc/policy.h; a
 * static string, valid for the life of the process). The knobs are read once per engine, at pe_create; the product needs
 * none of them (onnxruntime's session options are the reference's counterpart, src/cpp/piper.cpp:262-306). */
const char* pe_policy_describe(void);

/* In-process multi-GPU synthesis for C / C++ callers (the reference runs the phrases of a text one
 * after the other on one session, src/cpp/piper.cpp:549-582 -- they are independent, so they shard). One engine, one
 * stream and one worker thread per device. The voice is parsed and packed ONCE, on devices[0]; every other device gets an
 * identically laid out arena (pe_create_in_arena, skeleton) and receives the packed weights by ONE RCCL broadcast over
 * xGMI (ncclBroadcast on a communicator of the group's distinct devices; peer copies when librccl cannot be loaded) -- the
 * in-process counterpart of the torch.distributed broadcast in piper_amd/dist.py. A call deals its
 * utterances to the devices in longest-first order onto the least-loaded device (load = phoneme ids), runs the shards
 * concurrently and returns group-owned host views in the CALLER's order: sample_offsets / pcm / frames as in pe_result,
 * `audio` is NULL (fetch floats per engine if needed), infer_seconds = wall time of the whole call. The same device may
 * be listed more than once (engines sharing a GPU): a call then COALESCES that device's utterances onto as few of its
 * engines as 64-utterance shares need -- one batched call beats several single-utterance pipelines racing for the launch
 * path -- and pe_group_assignment reports which engines ran. Not thread-safe: one call at a time per group. */
typedef struct pe_group pe_group;
int pe_group_create(const void* blob, size_t nbytes, const int32_t* devices, int32_t n_devices, pe_group** out);
/* How the last pe_group_create on this thread moved the packed weights between devices: "rccl" (one ncclBroadcast on a
 * communicator of the group's distinct devices -- librccl is dlopen'ed on first use), "peer-copy (<why RCCL was not used>)",
 * "same-device" (all engines share one GPU) or "none" (one engine). PIPER_HIP_GROUP_BCAST=peer forces the copies, =rccl
 * takes the collective even for a single device (self-test on a one-GPU box). */
const char* pe_group_broadcast_path(void);
int32_t pe_group_size(pe_group* g);
pe_engine* pe_group_engine(pe_group* g, int32_t i);           /* e.g. pe_set_seed / pe_get_info / pe_profile_* per device */
int pe_group_synthesize_batch(pe_group* g, const int64_t* ids, const int64_t* offsets, int32_t batch,
                              const float scales[3], const int64_t* sids, pe_result* result);
/* which engine (index into the group) ran utterance i of the last call */
int pe_group_assignment(pe_group* g, int32_t* engine_index, int64_t capacity);
void pe_group_destroy(pe_group* g);

/* Concurrent single-utterance requests as batched engine calls (dynamic batching). The reference serves one phrase at a
 * time on one session (src/cpp/piper.cpp:549-582; its HTTP server, src/python_run/piper/http_server.py, one request at a
 * time); a server on this engine has many caller threads, each with ONE utterance. An engine handle is not thread-safe and a
 * B=1 pipeline leaves most of the chip idle, so: every thread calls pe_coalescer_synthesize (thread-safe, blocking); the
 * thread that finds the engine free leads -- it takes every request queued at that moment with the same scales (up to
 * max_batch; after waiting up to max_wait_us for stragglers, 0 = never wait) and runs them as ONE pe_synthesize_batch-style
 * call while later arrivals queue up for the next leader. Each request gets exactly what its own B=1 call computes: its
 * own noise draws, and int16 PCM peak-normalised over ITS waveform (piper.cpp:410-431) -- the batched kernels treat
 * utterances independently. *pcm is malloc'ed for the caller (pe_free). *batch_size = utterances of the engine call that
 * served the request. pe_coalescer_stats: engine calls / requests so far. The engine must outlive the coalescer and must
 * not be used directly while requests are in flight. */
typedef struct pe_coalescer pe_coalescer;
int pe_coalescer_create(pe_engine* e, int32_t max_batch, int32_t max_wait_us, pe_coalescer** out);
int pe_coalescer_synthesize(pe_coalescer* c, const int64_t* ids, int64_t n_ids, const float scales[3], int64_t sid,
                            int16_t** pcm, int64_t* n_samples, int32_t* frames, double* infer_seconds, int32_t* batch_size);
int pe_coalescer_stats(pe_coalescer* c, int64_t* engine_calls, int64_t* requests);
void pe_coalescer_destroy(pe_coalescer* c);

const char* pe_last_error(void);
void pe_destroy(pe_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* PIPER_HIP_H_ */
