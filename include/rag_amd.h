/*
 * rag_amd.h — C ABI of the MI355X-native retrieval hot path.
 *
 * The reference (IanHollow/rag-inference-pipeline) has no FFI of its own: its hot path is
 * three Python components that delegate to third-party native wheels.  This header is the
 * C-level seam those components bind instead of the wheels.  Each entry point cites the
 * reference call site it replaces (paths relative to the reference repo root):
 *
 *   rag_index_*      <- src/pipeline/components/faiss_store.py
 *                         load()    :40-111  (faiss.read_index + warm-up search)
 *                         search()  :113-158 (index.search(embeddings, k) -> (D, I))
 *                         unload()  :160-171, index_size :178-183
 *   rag_bert_*       <- src/pipeline/components/embedding.py   (query encoder)
 *                         load()    :70-98   (SentenceTransformer(name))
 *                         encode()  :100-175 (model.encode(..., normalize_embeddings=True))
 *                    <- src/pipeline/components/reranker.py    (cross-encoder)
 *                         load()    :71-173  (AutoModelForSequenceClassification)
 *                         rerank()  :206-272 (model(**inputs).logits -> sigmoid)
 *
 * Conventions
 *   - Every function returns an int status (RAG_OK == 0) unless documented otherwise; no C++
 *     exception crosses the boundary.  rag_last_error() returns a thread-local message.
 *   - Plain pointers and sizes only.  "host" pointers are ordinary process memory; "dev"
 *     pointers are HIP device memory on the handle's device.  `stream` is a hipStream_t
 *     passed as void*; work is enqueued on exactly that stream (NULL = HIP's default stream,
 *     which is also what torch.cuda.current_stream().cuda_stream reads as by default).  The
 *     host-pointer entry points use a private stream of the handle and block until done.
 *   - Handles are thread-safe: concurrent calls on one handle serialise on an internal mutex
 *     (the reference's BatchScheduler can have several batches in flight,
 *     services/gateway/batch_scheduler.py:286-288).
 *   - There is NO CPU fallback: without a usable HIP device every compute entry point fails
 *     with RAG_ERR_NO_DEVICE.
 */
#ifndef RAG_AMD_H
#define RAG_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped on EVERY change of a prototype, a struct layout or a constant below (rounds 1-3 forgot to: a library built from
 * an older header passed the loader's check).  The Python binding compares rag_abi_version() with THIS line, parsed from
 * the header it ships with, and rag_source_digest() with a digest of the csrc/ sources it ships with. */
#define RAG_AMD_ABI_VERSION 9

/* status codes */
#define RAG_OK 0
#define RAG_ERR_INVALID_ARG 1   /* bad pointer / size / k / metric */
#define RAG_ERR_NO_DEVICE 2     /* no HIP device, or device index out of range */
#define RAG_ERR_HIP 3           /* a HIP runtime call failed (see rag_last_error) */
#define RAG_ERR_OOM 4           /* device or pinned-host allocation failed */
#define RAG_ERR_UNSUPPORTED 5   /* shape outside what the kernels cover (d, k limits) */
#define RAG_ERR_STATE 6         /* call not valid in the handle's current state */

/* metrics: same meaning as faiss.METRIC_INNER_PRODUCT / faiss.METRIC_L2 (IndexFlatIP / IndexFlatL2) */
#define RAG_METRIC_INNER_PRODUCT 0
#define RAG_METRIC_L2 1

/* ---- library ------------------------------------------------------------------------- */

/* ABI version of the loaded library (compare with RAG_AMD_ABI_VERSION). Never fails. */
int rag_abi_version(void);

/* Digest of the sources this library was compiled from (hex SHA-256 over include/rag_amd.h and every file of csrc/ in
 * name order, passed in by the build as -DRAG_AMD_SOURCE_DIGEST; "unknown" for a hand build).  The loader refuses — or
 * rebuilds — a library whose digest differs from the sources beside it: *.so files are not tracked by git but do
 * travel with a working tree, so "the library is older than its callers" is a state that occurs.  Never fails. */
const char* rag_source_digest(void);

/* Number of visible HIP devices (0 when there is no GPU / no driver). Never fails. */
int rag_device_count(void);

/* Thread-local, NUL-terminated description of the last failure on this thread. */
const char* rag_last_error(void);

/* ---- flat index: exhaustive fp32 scan + exact top-k ----------------------------------- */

typedef struct rag_index rag_index;

/* Create an empty flat index of dimension d on `device`.
 * Replaces: faiss.read_index() producing an IndexFlatIP / IndexFlatL2 (faiss_store.py:66-69). */
int rag_index_create(int32_t d, int32_t metric, int32_t device, rag_index** out);

/* Release the index and its device memory.  Replaces FAISSStore.unload (faiss_store.py:160-171). */
int rag_index_destroy(rag_index* h);

/* Make room for at least n_total rows (optional; add grows geometrically otherwise). */
int rag_index_reserve(rag_index* h, int64_t n_total);

/* Append n rows (row-major, d floats each) from host memory; ids are insertion row numbers,
 * as IndexFlat assigns them (scripts/create_test_docs.py:92-97 relies on that). */
int rag_index_add(rag_index* h, const float* rows_host, int64_t n);

/* Same, rows already in device memory on the handle's device (device-to-device copy on `stream`). */
int rag_index_add_device(rag_index* h, const float* rows_dev, int64_t n, void* stream);

/* Append n deterministic synthetic unit-norm rows generated on the device (bench/test corpus;
 * the generator is restated bit-exactly by oracle/flat_oracle.c:rago_synth_rows). Row r of the
 * whole index is a pure function of (seed, global row number r + row_number_offset). */
int rag_index_add_synthetic(rag_index* h, int64_t n, uint64_t seed, int64_t row_number_offset);

/* Number of rows / dimension / metric.  Replaces index.ntotal (faiss_store.py:178-183). */
int64_t rag_index_ntotal(const rag_index* h);
int32_t rag_index_dim(const rag_index* h);
int32_t rag_index_metric(const rag_index* h);

/* Value added to every returned id (shard base row when the corpus is split over GPUs).
 * Global ids must fit 32 bits (the shard merge ranks 64-bit keys that carry the id in their low word):
 * 0 <= id_offset and id_offset + ntotal <= 2^32 - 1, else RAG_ERR_UNSUPPORTED — also from a later add. */
int rag_index_set_id_offset(rag_index* h, int64_t id_offset);

/* Exact search.  Replaces index.search(embeddings, k) (faiss_store.py:152).
 *   queries_host : nq x d fp32, row-major (caller-owned)
 *   out_scores   : nq x k fp32 — inner products sorted descending (IP) or squared L2
 *                  distances sorted ascending (L2)
 *   out_ids      : nq x k int64 — row ids; slots beyond ntotal hold -1 with score
 *                  -FLT_MAX (IP) / +FLT_MAX (L2), as IndexFlat pads them
 * Ties are broken by ascending id.  Blocks until the results are in the output buffers.
 *
 * Non-finite values.  Selection follows the heap faiss runs behind IndexFlat.search: it starts at
 * -FLT_MAX (IP) / +FLT_MAX (L2) and admits a candidate only if it compares strictly better.  Hence:
 *   - a row whose ranking score (IP: q.x; L2: 2 q.x - ||x||^2) is NaN, -inf or -FLT_MAX is never
 *     returned; if fewer than k rows remain, the tail is (-1, -FLT_MAX / +FLT_MAX) padding;
 *   - a query containing NaN therefore gets k padding slots; +inf scores rank first, ties by id;
 *   - L2: a query whose squared norm is inf or NaN gets k padding slots (every distance is inf or NaN);
 *   - products follow IEEE-754 (inf * 0 = NaN, inf - inf = NaN) in the canonical summation order;
 *   - the two-stage search (rag_index_set_screening) stands aside: a corpus with non-finite or
 *     extreme values reports RAG_SCREEN_INACTIVE, an out-of-range query is answered by the fp32 scan.
 * Held to oracle/flat_oracle.c by tests/test_flat_gpu.py::test_nonfinite_*. */
int rag_index_search(rag_index* h, const float* queries_host, int32_t nq, int32_t k,
                     float* out_scores, int64_t* out_ids);

/* Same with queries and outputs in device memory; asynchronous on `stream`. */
int rag_index_search_device(rag_index* h, const float* queries_dev, int32_t nq, int32_t k,
                            float* out_scores_dev, int64_t* out_ids_dev, void* stream);

/* Same search with the choice the two-stage mode leaves open made by the caller (rag_index_set_screening):
 *   RAG_SEARCH_DEFAULT         what rag_index_search_device does: a two-stage search enqueues its fp32 fallback
 *                              behind itself (two launches that switch themselves off on the device when no
 *                              certificate failed), so the call stays asynchronous and the result is always final;
 *   RAG_SEARCH_EXACT_ONE_PASS  never screen: the one-pass fp32 scan whatever the index's screening mode;
 *   RAG_SEARCH_DEFER_FALLBACK  a two-stage search WITHOUT its fallback launches: *flag_dev (a device word of the
 *                              caller's) is 0 when every query's certificate held — the result is final — and 1
 *                              when one failed: the caller then repeats the batch with RAG_SEARCH_EXACT_ONE_PASS.
 *                              The word is written on `stream` by the search's own kernels.  (In the other two
 *                              modes, and when the search was not screened at all, a non-null flag_dev is set to 0:
 *                              those results are final.)  This is what the sharded serving step uses: the
 *                              word travels in each rank's block of the all-gather, and the fp32 scan + a second
 *                              gather run only when some rank raised it (rag_inference_pipeline_amd/sharded.py).
 * No counterpart in the reference (faiss IndexFlat has one code path, faiss_store.py:152). */
#define RAG_SEARCH_DEFAULT 0
#define RAG_SEARCH_EXACT_ONE_PASS 1
#define RAG_SEARCH_DEFER_FALLBACK 2
int rag_index_search_device_ex(rag_index* h, const float* queries_dev, int32_t nq, int32_t k,
                               float* out_scores_dev, int64_t* out_ids_dev, int32_t mode, uint32_t* flag_dev,
                               void* stream);

/* Queries in device memory (e.g. left there by rag_bert_encode_to_device on the same stream), results in HOST
 * memory: the search is enqueued on `stream` behind whatever produced the queries, ids and scores come back
 * in one read-back each and the call blocks until they are in the output buffers.  This is the hand-off
 * between the embedder and the index inside one retrieval batch (services/retrieval/api.py:351-390 passes a
 * host array from one to the other; here the embeddings never leave HBM). */
int rag_index_search_device_host_out(rag_index* h, const float* queries_dev, int32_t nq, int32_t k,
                                     float* out_scores, int64_t* out_ids, void* stream);

/* Copy rows [row0, row0+n) back to host memory (parity spot checks at full size). */
int rag_index_get_rows(rag_index* h, int64_t row0, int64_t n, float* out_rows_host);

/* Scan-kernel profiling: when enabled every search brackets its scan kernel with HIP events
 * on the launch stream.  rag_index_profile returns the accumulated kernel time and launch
 * count since the last reset (it synchronises the recorded events). */
int rag_index_profile_enable(rag_index* h, int32_t on);
int rag_index_profile(rag_index* h, double* scan_ms_total, int64_t* scan_launches, int32_t reset);

/* A share of the chip for a stage (no counterpart in the reference, whose stages time-share one device):
 * rag_stream_create_masked returns a hipStream_t whose kernels run only on the compute units with mask bits
 * [first_cu, first_cu + n_cus) of hipExtStreamCreateWithCUMask's enumeration.  On MI355X that enumeration interleaves
 * the 8 XCDs and, inside an XCD, its 4 shader engines — 32 consecutive bits are one CU of every engine of every XCD
 * (scripts/exp/cumask_probe.hip).  Use multiples of 32: workgroups are dealt to the engines by count, so a share that
 * leaves the engines unequal puts two workgroups of a one-per-CU grid on one CU (measured: the corpus scan on 240, 232,
 * 216 or 208 CUs takes TWICE as long as on 224 or 256).  rag_index_set_cu_budget / rag_bert_set_cu_budget tell a handle how many CUs its
 * launches may count on (grid of the persistent scan kernel, tile and split-K choices of the small GEMMs); 0 = the
 * whole device (the default).  Use: the query encoder of batch i + 1 on 16 CUs beside the corpus scan of batch i on
 * the other 240 — the scan is HBM-bound and loses nothing, the encoder no longer shares CUs with it (bench.py,
 * with_query_encoder.pipelined).  Results do not depend on either setting. */
int rag_stream_create_masked(int32_t device, int32_t first_cu, int32_t n_cus, void** stream_out);
int rag_stream_destroy(int32_t device, void* stream);
/* Work enqueued on `waiter` after this call starts only when what is on `signaler` now has finished. */
int rag_stream_wait(int32_t device, void* waiter, void* signaler);
int rag_device_cu_count(int32_t device, int32_t* n_cus);
int rag_index_set_cu_budget(rag_index* h, int32_t n_cus);
/* (rag_bert_set_cu_budget is declared with the rag_bert entry points below.) */

/* Largest k the fused scan+select kernel supports for (d, nq) on this build; 0 if d unsupported. */
int32_t rag_index_max_k(int32_t d, int32_t nq);

/* Two-stage exact search (optional, off by default).  With RAG_SCREEN_FP16 the index keeps a scaled
 * fp16 copy of the corpus beside the fp32 rows (+50 % memory).  A search of k <= 100 then (1) scans
 * the copy — half the bytes of the fp32 scan — keeping every row whose approximate score lies within
 * a worst-case error band of the running k-th best, (2) recomputes the canonical fp32 score of those
 * candidates and ranks them, (3) checks per query a certificate that no row outside the candidate
 * list can reach the k-th exact score.  Queries that pass return exactly what the one-pass fp32
 * search returns (same ids, bit-identical scores); queries that fail are re-run through the fp32
 * scan on the device, so results never depend on the mode.  Applies to d <= 2048 (k <= 48 beyond
 * d = 1024, k <= 16 beyond 1536: the LDS budget) with finite corpus values of ordinary magnitude;
 * otherwise the index reports RAG_SCREEN_INACTIVE and searches use the fp32 scan.  (No counterpart in the reference: faiss IndexFlat has one code path.) */
#define RAG_SCREEN_OFF 0
#define RAG_SCREEN_FP16 1
#define RAG_SCREEN_INACTIVE 2   /* requested, but the corpus is outside what the error bound covers */
int rag_index_set_screening(rag_index* h, int32_t mode);
int32_t rag_index_screening(const rag_index* h);

/* Counters of the two-stage search since the last reset: queries answered, queries that fell back
 * to the fp32 scan, and the largest observed |approximate - exact| / (error bound) over all verified
 * candidates (must stay below 1; typically ~0.03). */
int rag_index_screen_stats(rag_index* h, int64_t* queries, int64_t* fallbacks, double* max_err_ratio,
                           int32_t reset);

/* Merge per-shard top-k lists (the step after the RCCL all-gather):
 *   scores_dev / ids_dev : n_shards x nq x k, each list sorted as rag_index_search returns it
 *   out                  : nq x k, same ordering rule (score, then ascending id)
 * metric picks the direction.  Asynchronous on `stream`. */
int rag_merge_topk_device(int32_t device, int32_t metric, int32_t n_shards, int32_t nq, int32_t k,
                          const float* scores_dev, const int64_t* ids_dev, float* out_scores_dev,
                          int64_t* out_ids_dev, void* stream);

/* Same, reading the all-gather's receive buffer in place: shard g's block starts at
 * packed_dev + g * shard_stride_bytes and holds nq*k int64 ids at offset 0 and nq*k fp32 scores at
 * scores_offset_bytes (the layout rag_inference_pipeline_amd/sharded.py:pack_layout produces).
 * Ids must be < 2^32 (the merge ranks on 64-bit (score, id) keys). */
int rag_merge_topk_packed_device(int32_t device, int32_t metric, int32_t n_shards, int32_t nq, int32_t k,
                                 const void* packed_dev, int64_t shard_stride_bytes,
                                 int64_t scores_offset_bytes, float* out_scores_dev, int64_t* out_ids_dev,
                                 void* stream);

/* Same, for blocks that also carry one 32-bit "my result is not final" word each (RAG_SEARCH_DEFER_FALLBACK) at
 * flag_offset_bytes: *any_flag_dev receives the OR of the n_shards words (written by the merge itself, so the
 * caller reads ONE word back, not one per rank).  host_mirror (NULL, or pinned host memory the device can write:
 * hipHostMalloc / torch pin_memory, 8-byte aligned, laid out like ONE shard's block — ids at 0, scores at
 * scores_offset_bytes, the OR-ed word at flag_offset_bytes) receives the same result from the merge kernel itself:
 * the batch then needs no read-back copy, only a wait for `stream`. */
int rag_merge_topk_packed_flagged_device(int32_t device, int32_t metric, int32_t n_shards, int32_t nq, int32_t k,
                                         const void* packed_dev, int64_t shard_stride_bytes,
                                         int64_t scores_offset_bytes, int64_t flag_offset_bytes,
                                         float* out_scores_dev, int64_t* out_ids_dev, uint32_t* any_flag_dev,
                                         void* host_mirror, void* stream);


/* ---- IVFFlat `nprobe` mode (optional) --------------------------------------------------------------------- */

/* The reference's data generator writes a faiss IndexIVFFlat (scripts/create_test_docs.py:83-104: L2, nlist 4096,
 * nprobe 64) and FAISSStore.load sets index.nprobe from the settings (faiss_store.py:84-92): on such a file
 * index.search (faiss_store.py:152) looks only at the rows of the `nprobe` inverted lists whose centroids are
 * nearest to the query.  rag_index_* answers such a file exhaustively (every list: more exact than the reference);
 * rag_ivf_* restates the reference's candidate set:
 *   1. coarse quantizer: the flat scan (rag_index machinery) over the nlist centroids, nprobe best per query under the
 *      quantizer's metric, ties by ascending list number;
 *   2. per pass of <= 32 queries the probe table becomes a mask of queries per list; every list that at least one query
 *      probes is read ONCE and multiplied with the pass's queries by the flat scan's MFMA loop (the flat search's
 *      canonical summation order and score bits); a query keeps candidates only from the lists it probes;
 *   3. the workgroups' per-query top-k lists are merged: (score, then ascending stored id).
 * faiss ranks equal scores by its heap's visiting order; here the rule is the flat search's.  With nprobe >= nlist the
 * result equals rag_index_search's bit for bit.  d <= 1024, nlist <= 32768, k <= 2048 (k above what one pass selects
 * — 112 at d = 768 — takes rounds, as in the flat search).  Parity status: as the flat search (unpinned against faiss
 * itself; held to oracle/flat.py:ivf_search). */
typedef struct rag_ivf rag_ivf;
int rag_ivf_create(int32_t d, int32_t metric, int32_t quantizer_metric, int32_t device, rag_ivf** out);
int rag_ivf_destroy(rag_ivf* h);

/* The trained index, once: nlist centroids (nlist x d fp32), the rows in LIST order (list l owns rows
 * list_offsets[l] .. list_offsets[l + 1] - 1 of rows_host / ids_host; list_offsets has nlist + 1 entries, from 0 to
 * ntotal) and each row's stored id (what a search returns; 0 <= id <= 2^32 - 2).  Host pointers; blocks. */
int rag_ivf_set_lists(rag_ivf* h, const float* centroids_host, int64_t nlist, const float* rows_host,
                      const int64_t* ids_host, const int64_t* list_offsets_host);
int64_t rag_ivf_ntotal(const rag_ivf* h);
int64_t rag_ivf_nlist(const rag_ivf* h);
/* Two-stage form of the list scan (as rag_index_set_screening for the flat search: a screening pass over a scaled fp16
 * copy of the lists — half the bytes — then canonical fp32 scores of the candidates under a per-query certificate, the
 * exact scan as the fallback; identical results).  On by default for k <= 100 when the corpus is inside the range the
 * error bound covers (+50 % index memory); RAG_AMD_IVF_TWO_STAGE=0 at load or at search time turns it off.
 * rag_ivf_two_stage: 1 when the copy exists and is valid.  rag_ivf_screen_stats: as rag_index_screen_stats. */
int32_t rag_ivf_two_stage(const rag_ivf* h);
int rag_ivf_screen_stats(rag_ivf* h, int64_t* queries, int64_t* fallbacks, double* max_err_ratio, int32_t reset);

/* index.search(embeddings, k) with index.nprobe = nprobe (clamped to nlist).  Same output conventions as
 * rag_index_search (-1 / -+FLT_MAX padding when fewer than k rows are reachable).  Host pointers; blocks. */
int rag_ivf_search(rag_ivf* h, const float* queries_host, int32_t nq, int32_t k, int32_t nprobe, float* out_scores,
                   int64_t* out_ids);
/* The same with device pointers, enqueued on `stream` (the caller's; NULL = the default stream) without a host
 * round trip — the form the embedder's device hand-off uses (rag_index_search_device's contract). */
int rag_ivf_search_device(rag_ivf* h, const float* queries_dev, int32_t nq, int32_t k, int32_t nprobe,
                          float* out_scores_dev, int64_t* out_ids_dev, void* stream);
/* Device queries in (an embedder's result, still on `stream`), host results out; blocks until they are there
 * (rag_index_search_device_host_out's contract). */
int rag_ivf_search_device_host_out(rag_ivf* h, const float* queries_dev, int32_t nq, int32_t k, int32_t nprobe,
                                   float* out_scores, int64_t* out_ids, void* stream);

/* ---- C1: the shard step's collectives on an own RCCL communicator (SURVEY §8a "C1", §8e) ------------------- */

/* The reference has no multi-device code (faiss_store.py:37: one index, one process).  With the corpus split over G
 * GPUs — one process each — a search is: local scan + top-k -> ONE all-gather of every rank's packed [ids | scores |
 * flag] block -> merge of the G lists on every rank.  These entry points put the collective on the SAME stream as the
 * kernels around it: torch.distributed (or any launcher) only carries the 128-byte unique id from rank 0 to the other
 * ranks; it is not on the data path.  RCCL is bound at run time (see csrc/rag_comm.hip for the search order). */
typedef struct rag_comm rag_comm;
#define RAG_COMM_ID_BYTES 128
#define RAG_COMM_HEAD_WORDS 4   /* a request's head: four 64-bit words [op, nq, k, d] (rag_inference_pipeline_amd/sharded.py) */

/* Bind RCCL (NULL: default search order) and report its version code (e.g. 22606); idempotent. */
int rag_comm_runtime(const char* librccl_path, int32_t* version_out);

/* Rank 0: a fresh unique id (ncclGetUniqueId), to be handed to every rank by the launcher's own means. */
int rag_comm_unique_id(uint8_t* id_out /* RAG_COMM_ID_BYTES */);

/* Collective: every rank of the group calls it with the same id (ncclCommInitRank on `device`). */
int rag_comm_create(const uint8_t* id, int32_t rank, int32_t world, int32_t device, rag_comm** out);
int rag_comm_destroy(rag_comm* c);
int32_t rag_comm_rank(const rag_comm* c);
int32_t rag_comm_world(const rag_comm* c);

/* ncclAllGather / ncclBroadcast of raw bytes, asynchronous on `stream` (device pointers on the communicator's device;
 * recv_dev holds world * bytes_per_rank bytes, rank r's block at r * bytes_per_rank). */
int rag_comm_all_gather_device(rag_comm* c, const void* send_dev, void* recv_dev, int64_t bytes_per_rank, void* stream);
int rag_comm_broadcast_device(rag_comm* c, void* buf_dev, int64_t bytes, int32_t root, void* stream);

/* One served request = one fixed-size message [RAG_COMM_HEAD_WORDS x int64 head | payload] (the query batch).  On
 * `root` the message is uploaded from msg_host (pinned; may be NULL when msg_dev already holds it) and broadcast; on
 * every rank a last, tiny kernel stores the head and then `seq` (> 0, increasing) into head_mirror — pinned host
 * memory of RAG_COMM_HEAD_WORDS + 1 64-bit words the device can write (may be NULL) — so that a follower learns what to
 * launch by polling one host word (rag_comm_wait_head) instead of synchronising the stream.  Asynchronous on `stream`;
 * the queries stay in msg_dev for the search that follows on the same stream. */
int rag_comm_request_device(rag_comm* c, const void* msg_host, void* msg_dev, int64_t bytes, int32_t root,
                            void* head_mirror, uint64_t seq, void* stream);

/* Spin (then nap in 50 us steps once 2 ms have passed) until head_mirror's sequence word equals `seq`; copies the
 * head's RAG_COMM_HEAD_WORDS words to head_out.  timeout_us < 0: wait for ever.  RAG_ERR_STATE on timeout. */
int rag_comm_wait_head(const void* head_mirror, uint64_t seq, int64_t timeout_us, int64_t* head_out);

/* Byte layout of one rank's packed result block for (nq, k): nq*k int64 ids at 0, nq*k fp32 scores at
 * *scores_offset, one uint32 "not final" word at *flag_offset, *block_bytes rounded up to a multiple of 8. */
int rag_pack_layout(int32_t nq, int32_t k, int64_t* scores_offset, int64_t* flag_offset, int64_t* block_bytes);

/* The whole shard step in one call, enqueued on ONE stream:
 *   rag_index_search_device_ex(h, queries_dev, nq, k, ..., mode, flag) into pack_dev (this rank's block, rag_pack_layout)
 *   -> ncclAllGather(pack_dev -> gathered_dev: world blocks)
 *   -> rag_merge_topk_packed_flagged_device(gathered_dev) into out_scores_dev / out_ids_dev / any_flag_dev and, when
 *      host_mirror is given (pinned, one block's layout), into host memory by the merge kernel itself.
 * comm_stream (NULL: `stream`) may name a second stream for the all-gather and the merge: they then start when the
 * local search has finished on `stream` (an event) and run BESIDE the next batch's local search — over xGMI the
 * collective's latency leaves the step time.  The caller waits for comm_stream (or `stream`) before reading results.
 * Replaces, for a sharded index, what faiss_store.py:152 does in one process. */
int rag_index_search_gather_device(rag_index* h, rag_comm* c, const float* queries_dev, int32_t nq, int32_t k,
                                   int32_t mode, void* pack_dev, void* gathered_dev, float* out_scores_dev,
                                   int64_t* out_ids_dev, uint32_t* any_flag_dev, void* host_mirror, void* stream,
                                   void* comm_stream);
/* The same step for a rank of a sharded IVFFlat index in the nprobe mode (every rank holds its share of every list): the
 * local search is rag_ivf_search_device(h, ..., nprobe, ...); its list is always final (the two-stage form resolves its
 * fallback on the device), so the block's flag word is 0. */
int rag_ivf_search_gather_device(rag_ivf* h, rag_comm* c, const float* queries_dev, int32_t nq, int32_t k, int32_t nprobe,
                                 void* pack_dev, void* gathered_dev, float* out_scores_dev, int64_t* out_ids_dev,
                                 uint32_t* any_flag_dev, void* host_mirror, void* stream, void* comm_stream);

/* ---- BERT-family transformer: query encoder and cross-encoder ----------------------------- */

typedef struct rag_bert rag_bert;

/* activations */
#define RAG_ACT_GELU 1       /* exact (erf) GELU — BERT / MiniLM / bge / XLM-R default */
#define RAG_ACT_GELU_TANH 2
#define RAG_ACT_RELU 3

/* classifier heads (for RAG_BERT_OUT_LOGITS) */
#define RAG_HEAD_NONE 0
#define RAG_HEAD_BERT 1      /* BertForSequenceClassification: tanh(Wp x_cls + bp) -> Wc . + bc */
#define RAG_HEAD_ROBERTA 2   /* (XLM-)RobertaForSequenceClassification: tanh(Wd x_cls + bd) -> Wo . + bo */

/* outputs of a forward pass */
#define RAG_BERT_OUT_MEAN 0    /* nseq x hidden: mean over tokens (sentence-transformers Pooling "mean") */
#define RAG_BERT_OUT_CLS 1     /* nseq x hidden: first token (Pooling "cls") */
#define RAG_BERT_OUT_LOGITS 2  /* nseq x n_labels classifier logits */
#define RAG_BERT_OUT_PROBS 3   /* nseq x n_labels sigmoid(logits) (reranker.py:252) */
#define RAG_BERT_OUT_HIDDEN 4  /* T x hidden last hidden state (tests) */

typedef struct rag_bert_config {
    int32_t vocab_size;
    int32_t hidden;        /* <= 1024, multiple of 32 */
    int32_t n_layers;
    int32_t n_heads;       /* hidden / n_heads must be 32 or 64 */
    int32_t intermediate;  /* multiple of 32 */
    int32_t max_positions;
    int32_t type_vocab;    /* 0 = no token-type embedding */
    int32_t pos_offset;    /* 0 for BERT; padding_idx + 1 (= 2) for RoBERTa-family position ids */
    int32_t act;           /* RAG_ACT_* */
    int32_t head;          /* RAG_HEAD_* */
    int32_t n_labels;      /* classifier outputs (1 for the rerankers) */
    float ln_eps;
    int32_t gemm_mode;     /* RAG_GEMM_*: how big-batch (> 1024 tokens) GEMMs run */
} rag_bert_config;

/* RAG_GEMM_F32 (default): fp32 results.  The layers' GEMMs run on the fp16 matrix cores with every fp32 operand
 *   written as TWO fp16 numbers — hi = fp16(x), lo = fp16((x - hi) 2^11): 22 significand bits — and the three products
 *   hi hi, lo hi, hi lo accumulated in fp32 (the cross terms in their own accumulator, scaled by 2^-11 at the end): the
 *   error stays below the fp32 accumulation's own rounding (measured against float64: at or below the fp32-MFMA
 *   path's), at 5x the fp32 MFMA rate.  fp16's exponent range is narrower than fp32's: a pass in which an activation
 *   reaches |x| >= 65504 raises a flag on the device, and the host-pointer entry point then repeats that pass with
 *   every operand split exactly into three bf16 numbers (six products; fp32's range; built on first need) — so results
 *   never depend on the range, only the time does (rag_bert_range_events counts such passes).  A checkpoint with a
 *   GEMM weight outside fp16's range uses the three-plane path throughout.  The library builds the split weight
 *   images at rag_bert_create (+1x the GEMM weights' memory).
 * RAG_GEMM_F16: the precision the reference runs its reranker at on a GPU (reranker.py:91-93, model.half()).  Above
 *   1024 tokens the whole pass keeps its activations in fp16, stored in MFMA-fragment order (csrc/gemm_wt.hip.h): GEMMs
 *   and attention on the fp16 matrix cores with fp32 accumulation, softmax and LayerNorm statistics in fp32, bias /
 *   activation / residual joined to the fp32 sums before the one rounding to fp16.  Outputs stay fp32 in the caller's
 *   layout.  The library builds the fp16 weight image itself.  Batches of 1024 tokens or fewer run as in
 *   RAG_GEMM_F32_STRICT.
 * RAG_GEMM_F32_STRICT: every GEMM on the fp32 MFMA (v_mfma_f32_32x32x2_f32 chains). */
#define RAG_GEMM_F32 0
#define RAG_GEMM_F16 1
#define RAG_GEMM_F32_STRICT 2

/* Number of entries rag_bert_create expects in `weights` for a config:
 *   [0] word_emb [V][H]  [1] pos_emb [P][H]  [2] type_emb [Tv][H] (NULL if type_vocab == 0)
 *   [3] emb_ln_gamma [H] [4] emb_ln_beta [H]
 *   per layer l, 12 entries from 5 + 12 l:
 *     qkv_w [3H][H] (q;k;v rows) qkv_b [3H]  attn_out_w [H][H] attn_out_b [H]  ln1_gamma ln1_beta
 *     ffn_in_w [I][H] ffn_in_b [I]  ffn_out_w [H][I] ffn_out_b [H]  ln2_gamma ln2_beta
 *   then, if head != RAG_HEAD_NONE: head_dense_w [H][H] head_dense_b [H] head_out_w [n_labels][H] head_out_b
 * All fp32, torch.nn.Linear layout (W[out][in], row-major), device memory on `device`.  The library
 * does not copy them: the caller (PyTorch-ROCm tensors) keeps them alive until rag_bert_destroy. */
int32_t rag_bert_weight_count(const rag_bert_config* cfg);

int rag_bert_create(const rag_bert_config* cfg, const void* const* weights_dev, int32_t n_weights,
                    int32_t device, rag_bert** out);
int rag_bert_destroy(rag_bert* h);

/* Forward pass over nseq PACKED sequences (no padding): sequence s owns tokens
 * cu_seqlens[s] .. cu_seqlens[s+1]-1 of ids / type_ids (type_ids may be NULL = all zero).
 * `out_kind` is RAG_BERT_OUT_*; `normalize` L2-normalises pooled embeddings
 * (normalize_embeddings=True, embedding.py:132).  Host buffers; blocks until `out` is filled. */
int rag_bert_forward(rag_bert* h, const int32_t* ids, const int32_t* type_ids, const int32_t* cu_seqlens,
                     int32_t nseq, int32_t out_kind, int32_t normalize, float* out);

/* Same with ids / type_ids / cu_seqlens / out in device memory (total_tokens = cu_seqlens[nseq],
 * max_seq_len = longest sequence); asynchronous on `stream`.  range_flag (RAG_GEMM_F32; NULL = the handle's own word,
 * see rag_bert_range_events): a 32-bit word in PINNED HOST memory the device can write (hipHostMalloc / torch
 * pin_memory), zeroed by the caller — the GEMM kernels store 1 into it when an activation of this pass is outside
 * fp16's range (|x| >= 65504), in which case the pass's output is void and the caller repeats it through the
 * host-pointer entry point (which falls back to fp32's range by itself).  Read it once `stream` has passed the pass. */
int rag_bert_forward_device(rag_bert* h, const int32_t* ids_dev, const int32_t* type_ids_dev,
                            const int32_t* cu_seqlens_dev, int32_t nseq, int32_t total_tokens,
                            int32_t max_seq_len, int32_t out_kind, int32_t normalize, float* out_dev,
                            uint32_t* range_flag, void* stream);

/* Background mode (off by default): small-batch GEMMs (<= 1024 tokens: the query encoder) run in a form that needs
 * 32 KiB of LDS and four waves per workgroup instead of 96-144 KiB and sixteen.  For a caller that runs the encoder on
 * a side stream UNDER a long-running kernel of another stream — the corpus scan keeps a workgroup with 65-115 KiB of
 * LDS resident on every CU for milliseconds, and only workgroups that fit beside it start before it ends.  Slower
 * when nothing else runs (a barrier per 16-deep K-step), same results.  (No counterpart in the reference.) */
int rag_bert_set_background(rag_bert* h, int32_t on);

/* CUs this model's launches may count on; 0 = the whole device (see rag_stream_create_masked). */
int rag_bert_set_cu_budget(rag_bert* h, int32_t n_cus);

/* The stream the host-ids entry points (rag_bert_forward, rag_bert_forward_to_device) enqueue on, instead of the
 * handle's own; NULL restores that.  The caller owns it and keeps it alive until it is replaced or the handle destroyed
 * (components/embedding.py: a stream restricted to the encoder's share of the chip, RAG_AMD_ENCODER_CUS). */
int rag_bert_set_stream(rag_bert* h, void* stream);

/* Passes the host-pointer entry point has repeated on the three-plane path because an activation left fp16's range
 * (RAG_GEMM_F32, see above).  `pending` (may be NULL; synchronises the device when given): 1 when a pass of one of the
 * ASYNCHRONOUS entry points (rag_bert_forward_device / _to_device) has raised the flag since it was last taken — those
 * cannot repeat a pass themselves, their caller must (taking the flag clears it). */
int rag_bert_range_events(rag_bert* h, int64_t* repeated_passes, int32_t* pending);

/* Token ids in HOST memory, result left in DEVICE memory: the hand-off from the embedder to the index inside a
 * retrieval batch (services/retrieval/api.py:351-390 passes a host array between them; here the embeddings stay
 * in HBM).  The ids are staged through pinned memory and uploaded on the handle's private stream, the forward
 * pass is enqueued behind them, and the call returns WITHOUT waiting: *stream_out is that stream (a hipStream_t),
 * on which the consumer enqueues its own work (rag_index_search_device_host_out) or which it synchronises.
 * Batches of up to 71 sequences / ~1000 tokens with a pooled output (the query encoder) are enqueued as ONE hipGraph
 * replay instead of 45-90 kernel launches: the shape is padded to a bucket with dummy sequences, every address in the
 * graph is fixed, and where the result and the range flag go travels through device cells that the graph's last node
 * reads (graphs are cached per padded shape, least recently used first out).  Halves the call's host time; results
 * equal the eager path's to fp32 rounding (the padded token count can change a split-K choice).
 * RAG_AMD_ENCODER_GRAPH=0 at rag_bert_create turns it off.
 * out_dev is caller-owned device memory (nseq x hidden / n_labels floats, by out_kind) and must stay allocated
 * until that stream has passed this call's work.  range_flag: as for rag_bert_forward_device. */
int rag_bert_forward_to_device(rag_bert* h, const int32_t* ids, const int32_t* type_ids,
                               const int32_t* cu_seqlens, int32_t nseq, int32_t out_kind, int32_t normalize,
                               float* out_dev, uint32_t* range_flag, void** stream_out);

/* ---- `compressed` document payload (host side) ------------------------------------------- */

/* LZ4 block compression of src[0, n) into dst (capacity cap >= rag_lz4_compress_bound(n) always
 * suffices); returns the compressed size, or -1 if dst is too small.  Python wraps blocks in the LZ4
 * frame format (rag_inference_pipeline_amd/lz4frame.py) — what the reference produces with
 * lz4.frame.compress(msgspec.json.encode(docs)) (services/retrieval/api.py:516-523) and the
 * generation node reads with lz4.frame.decompress (services/generation/service.py:429). */
int64_t rag_lz4_compress_bound(int64_t n);
int64_t rag_lz4_block_compress(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap);

/* xxHash32 (the LZ4 frame header checksum byte is (xxh32(descriptor, 0) >> 8) & 0xFF). */
uint32_t rag_xxh32(const uint8_t* data, int64_t n, uint32_t seed);

/* ---- stand-in tokenizer of the synthetic models (host side) --------------------------------------- */

/* (query, document) pairs -> the packed arrays rag_bert_forward takes, for ASCII text and the stand-in vocabulary of
 * models without a checkpoint (model_source.HashTokenizer: lower-cased word / punctuation pieces, id = first_id +
 * crc32(piece) % span): [cls] a [sep] b [sep] with token types 0 / 1, or <s> a </s></s> b </s> when roberta != 0,
 * truncated longest-first to max_length — what the reference asks of its tokenizer (reranker.py:237-246).
 * ids_out / types_out (may be NULL) hold `cap` entries, cu_out n_pairs + 1.  Returns the total number of tokens;
 * -1 if a text holds a non-ASCII byte (the caller falls back to its own regex), -2 if cap is too small, -3 on bad
 * arguments. */
int64_t rag_hash_encode_pairs(const uint8_t* const* first, const int64_t* first_len, const uint8_t* const* second,
                              const int64_t* second_len, int64_t n_pairs, int32_t max_length, int32_t roberta,
                              int32_t first_id, int32_t span, int32_t cls_id, int32_t sep_id, int32_t* ids_out,
                              int32_t* types_out, int32_t* cu_out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* RAG_AMD_H */
