/*
 * vaporetto_hip.h -- C ABI of the MI355X-native Vaporetto boundary scorer (libvaporetto_hip.so).
 *
 * The reference (daac-tools/vaporetto) has no FFI: its boundary is the public Rust API of the `vaporetto`
 * crate (/root/reference/vaporetto/src/lib.rs:82-91).  Each entry point below names the reference item it
 * stands in for; INTEGRATION.md shows the Rust `extern "C"` block and the `Predictor`/`Sentence` shim a
 * maintainer would add on top.  Plain pointers and sizes only; no torch, no C++ types.
 *
 * Conventions
 *   - every function returns a vpt_status; vpt_last_error() gives the message of the calling thread's last
 *     failure (VaporettoError's Display text where the reference defines one, errors.rs:15-38);
 *   - a vpt_predictor is immutable after creation and may be shared by any number of host threads
 *     (mirrors `&self` in Predictor::predict, predictor.rs:518); all mutable state lives in the caller's
 *     buffers or in a per-caller vpt_batch workspace (mirrors the caller-owned `Sentence`);
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails with
 *     VPT_RUNTIME_ERROR.
 */
#ifndef VAPORETTO_HIP_H
#define VAPORETTO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vpt_status {
    VPT_OK = 0,
    VPT_INVALID_MODEL = 1,    /* VaporettoError::InvalidModel / DecodeError   (errors.rs:15-38) */
    VPT_INVALID_ARGUMENT = 2, /* VaporettoError::InvalidArgument              (errors.rs:15-38) */
    VPT_RUNTIME_ERROR = 3     /* HIP / device failure, or no device */
} vpt_status;

/* CharacterBoundary discriminants written to `labels` (sentence.rs:70-82) */
enum { VPT_NOT_WORD_BOUNDARY = 0, VPT_WORD_BOUNDARY = 1, VPT_BOUNDARY_UNKNOWN = 2 };

/* Flags of the *_flags entry points and of vpt_batch_set_flags.
 * VPT_FLAG_KYTEA_FULLWIDTH: score the text as KyteaFullwidthFilter would rewrite it
 *   (vaporetto_rules/src/string_filters/kytea_fullwidth.rs:13-117: a 1:1 char map, so boundaries keep their places);
 *   this is what the reference's CLI does before update_raw unless --no-norm is given (predict/src/main.rs:126-129),
 *   folded into the kernel's char classification table at no extra cost. */
enum { VPT_FLAG_KYTEA_FULLWIDTH = 1 };
/* Post-filters on the LABELS (scores are never touched), applied in this order after the sign threshold:
 * VPT_FLAG_WSCONST(t), t = CharacterType 1..6: KyteaWsConstFilter::new(t) -- a boundary between two chars of type t
 *   becomes NotWordBoundary (vaporetto_rules/src/sentence_filters/kytea_wsconst.rs:26-43; the CLI's --wsconst);
 * VPT_FLAG_SPLIT_LINEBREAKS: SplitLinebreaksFilter -- a boundary next to '\r' or '\n' becomes WordBoundary
 *   (vaporetto_rules/src/sentence_filters/split_linebreaks.rs:9-36).
 * Types are those of the scored text (after VPT_FLAG_KYTEA_FULLWIDTH, as in the CLI). */
#define VPT_FLAG_WSCONST(char_type) (1u << (char_type))
enum { VPT_FLAG_SPLIT_LINEBREAKS = 1 << 7, VPT_FLAG_ALL = 0xFF };

typedef struct vpt_predictor vpt_predictor;
typedef struct vpt_batch vpt_batch;

/* Message of the calling thread's most recent failed call ("" if none). Never NULL. */
const char *vpt_last_error(void);

/* Library version string. */
const char *vpt_version(void);

/* ---------------------------------------------------------------------------------------------------------
 * Model::read_slice + Predictor::new            (model.rs:127-135, predictor.rs:450-508)
 *
 * model_bytes : an un-compressed model file: "VaporettoTokenizer 0.5.0\n" + bincode (zstd is outside the
 *               API in the reference too, README.md:50-63).  The bytes are copied; the caller may free them.
 * predict_tags: as Predictor::new's flag.  Tag models are parsed and validated; boundary scores are
 *               identical either way; tag scoring itself is vpt_fill_tags_batch below.
 * device_id   : HIP device ordinal.
 * Errors      : VPT_INVALID_MODEL  "model version mismatch" | decode error | "failed to build the automaton"
 *               (empty pattern) | "invalid character type n-grams" (empty/duplicate type n-gram, cache
 *               variant, boundary_scorer_cache.rs:23-24) | "words must be shorter than or equal to 32767
 *               characters" | weight vector longer than its pattern allows | a tag n-gram whose rel_position exceeds
 *               the window (model contract);
 *               VPT_RUNTIME_ERROR when the device cannot be used.
 */
vpt_status vpt_predictor_create(const uint8_t *model_bytes, size_t len, int predict_tags, int device_id,
                                vpt_predictor **out);
void vpt_predictor_destroy(vpt_predictor *p);

/* ---------------------------------------------------------------------------------------------------------
 * The compiled predictor: Predictor::serialize_to_vec / deserialize_from_slice_unchecked   (predictor.rs:640-664)
 *
 * Compiling a bccwj-suw+unidic-sized model takes seconds; the result -- every device table plus a fixed-size
 * description -- can be saved and loaded back without the model (a format of this library: not the reference's
 * daachorse-internal one, and only valid for the library version that wrote it; like the reference's *_unchecked
 * loader it trusts the bytes beyond a version check and a checksum), and copied from one GPU to another device to
 * device (xGMI between the GPUs of a node), which is how the ranks of a multi-GPU job get their predictor.
 *
 * vpt_predictor_save : *needed = the size of the compiled form; with out == NULL only that; else capacity must cover it.
 * vpt_predictor_load : VPT_INVALID_MODEL "not a compiled predictor" | "... version mismatch" | "... truncated" |
 *                      "... corrupt (checksum)".
 * vpt_predictor_clone_to_device: a predictor on device_id with the same tables (hipMemcpyPeer; same device: a copy). */
vpt_status vpt_predictor_save(const vpt_predictor *p, uint8_t *out, size_t capacity, size_t *needed);
vpt_status vpt_predictor_load(const uint8_t *blob, size_t len, int device_id, vpt_predictor **out);
vpt_status vpt_predictor_clone_to_device(const vpt_predictor *src, int device_id, vpt_predictor **out);
/* For one process per GPU: vpt_predictor_describe gives the fixed-size description (meta_out may be NULL to ask for its size)
 * and the device address and size of the table arena; the caller moves those bytes to the other ranks' GPUs itself -- one
 * RCCL broadcast over xGMI -- and every rank turns what it received into a predictor with vpt_predictor_adopt_device
 * (device-to-device copy into an allocation the predictor owns; d_arena stays the caller's). */
vpt_status vpt_predictor_describe(const vpt_predictor *p, uint8_t *meta_out, size_t capacity, size_t *meta_bytes,
                                  const void **d_arena, size_t *arena_bytes);
vpt_status vpt_predictor_adopt_device(const uint8_t *meta, size_t meta_len, const void *d_arena, size_t arena_bytes,
                                      int device_id, vpt_predictor **out);

/* ---------------------------------------------------------------------------------------------------------
 * Sentence::from_raw's bookkeeping for a batch     (sentence.rs:160-196)
 *
 * Sentence i is utf8[byte_offsets[i] .. byte_offsets[i+1]) (valid UTF-8, as a Rust &str always is).
 * Writes out_offsets[0..S]: out_offsets[i] = sum_{j<i} (chars(j) - 1), i.e. where sentence i's n-1 boundary
 * scores start in the flat output arrays.  Host-only, no device needed.
 * Errors: VPT_INVALID_ARGUMENT "text: must contain at least one character" | "text: must not contain NULL".
 */
vpt_status vpt_count_boundaries(const uint8_t *utf8, const uint64_t *byte_offsets, size_t n_sentences,
                                uint64_t *out_offsets);

/* ---------------------------------------------------------------------------------------------------------
 * Predictor::predict over a batch of sentences, host buffers     (predictor.rs:518-543)
 *
 * scores_out[out_offsets[i] + b] = boundary score b of sentence i (what Sentence::boundary_scores() returns,
 * sentence.rs:1040-1046); labels_out likewise (Sentence::boundaries(), sentence.rs:993).  Either output
 * pointer may be NULL.  out_offsets must come from vpt_count_boundaries.  Copies H2D, launches, copies D2H,
 * synchronises.  Thread-safe on a shared predictor.
 */
vpt_status vpt_predict_batch(const vpt_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets,
                             size_t n_sentences, int32_t *scores_out, uint8_t *labels_out,
                             const uint64_t *out_offsets);

/* The same with VPT_FLAG_* (0 = exactly vpt_predict_batch). */
vpt_status vpt_predict_batch_flags(const vpt_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets,
                                   size_t n_sentences, int32_t *scores_out, uint8_t *labels_out,
                                   const uint64_t *out_offsets, unsigned flags);

/* Host buffers and the PCIe link.  vpt_predict_batch cuts a large batch into chunks and runs the copy in of one chunk, the
 * kernels of another and the copy out of a third at the same time: up to 16 M chars as 512 K-char chunks alternating over four
 * streams (a chunk's copy in, kernels and copy out in order on one of them), larger batches as 4 M-char chunks on a copy-in, a
 * compute and a copy-out stream.  Either output pointer may be NULL: a caller that only needs the labels saves four of the
 * five bytes per boundary that cross the link back.  With PINNED caller buffers -- vpt_host_alloc, or memory the caller registered with HIP
 * itself -- the copies are DMA transfers in both directions at once; with pageable buffers they go through the runtime's
 * staging.  vpt_host_alloc / vpt_host_free = hipHostMalloc / hipHostFree, exported so that callers need no HIP headers. */
vpt_status vpt_host_alloc(size_t bytes, void **out);
void vpt_host_free(void *ptr);

/* Predictor::predict over ONE batch on SEVERAL GPUs of a node: the batch is cut into n_preds contiguous sentence ranges of
 * about equal CHARACTER count (vpt_shard_bounds), range r is scored by preds[r] (normally vpt_predictor_clone_to_device
 * copies of one predictor) on a host thread of its own and written straight into its slice of scores_out / labels_out.
 * Sentences are independent (predictor.rs:518-543 keeps no state between calls), so there is no exchange step.
 * Arguments as for vpt_predict_batch_flags.  Mirrors how vaporetto_tantivy shares one Arc<Predictor> between indexing
 * threads (vaporetto_tantivy/src/lib.rs:62-67), with one device per thread. */
vpt_status vpt_predict_batch_sharded(const vpt_predictor *const *preds, size_t n_preds, const uint8_t *utf8,
                                     const uint64_t *byte_offsets, size_t n_sentences, int32_t *scores_out,
                                     uint8_t *labels_out, const uint64_t *out_offsets, unsigned flags);
/* bounds[0 .. n_shards]: shard r = sentences bounds[r] .. bounds[r+1], balanced by chars (host only). */
vpt_status vpt_shard_bounds(const uint64_t *out_offsets, size_t n_sentences, size_t n_shards, uint64_t *bounds);

/* Sentence::char_types for a batch (sentence.rs:1016; CharacterType::get_type, sentence.rs:50-67), computed on the device:
 * types_out[out_offsets[i] + i + c] = CharacterType (1..6) of char c of sentence i; flags: VPT_FLAG_KYTEA_FULLWIDTH gives the
 * types of the normalised text.  The *_device variant: device pointers, asynchronous, flags from vpt_batch_set_flags. */
vpt_status vpt_char_types_batch(const vpt_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets,
                                size_t n_sentences, const uint64_t *out_offsets, unsigned flags, uint8_t *types_out);
vpt_status vpt_char_types_batch_device(const vpt_predictor *p, vpt_batch *b, const uint8_t *d_utf8,
                                       const uint64_t *d_byte_offsets, const uint64_t *d_out_offsets,
                                       size_t n_sentences, uint64_t total_boundaries, uint8_t *d_types_out,
                                       void *hip_stream);

/* Sentence::from_raw + Predictor::predict for one sentence (same path, batch of one).
 * scores/labels need room for chars-1 entries (<= len-1). */
vpt_status vpt_predict_one(const vpt_predictor *p, const uint8_t *utf8, size_t len, int32_t *scores,
                           uint8_t *labels, size_t *n_boundaries);

/* ---------------------------------------------------------------------------------------------------------
 * Device-resident variant: inputs and outputs already in HBM, asynchronous on a HIP stream.
 *
 * A vpt_batch is the per-caller mutable workspace (what a reused `Sentence` is to the reference's callers,
 * predict/src/main.rs:122,129): tile tables, the device error word, timing events.  One per host thread /
 * stream; not shareable between concurrent calls.
 */
vpt_status vpt_batch_create(const vpt_predictor *p, vpt_batch **out);
void vpt_batch_destroy(vpt_batch *b);

/* d_utf8, d_byte_offsets[S+1], d_out_offsets[S+1], d_scores, d_labels are device pointers (d_scores or
 * d_labels may be NULL).  total_boundaries = out_offsets[S] (the caller sized the outputs with it), or any upper bound of it
 * when the offsets were made on the device and the caller will not wait for them (text bytes - S always is one; the same bound
 * must then be passed to the fill_tags / write calls that follow on this batch).
 * max_sentence_bytes: an upper bound on the byte length of any sentence (sizes the long-sentence scratch).
 * hip_stream: a hipStream_t (NULL = default stream).  Returns after enqueueing; errors found on the device
 * (empty sentence, NUL char, offsets inconsistent with the text) are reported by vpt_batch_sync. */
vpt_status vpt_predict_batch_device(const vpt_predictor *p, vpt_batch *b, const uint8_t *d_utf8,
                                    const uint64_t *d_byte_offsets, const uint64_t *d_out_offsets,
                                    size_t n_sentences, uint64_t total_boundaries, uint64_t max_sentence_bytes,
                                    int32_t *d_scores, uint8_t *d_labels, void *hip_stream);

/* Optional hint for the following vpt_predict_batch_device calls on this workspace: an upper bound on the number of
 * CHARACTERS of any sentence (0 = unknown: max_sentence_bytes is used, which is 3x pessimistic for Japanese text
 * and leaves the kernel's tiles about 10 % emptier).  An understated bound is reported by vpt_batch_sync like an
 * understated max_sentence_bytes. */
vpt_status vpt_batch_set_max_sentence_chars(vpt_batch *b, uint64_t max_sentence_chars);

/* VPT_FLAG_* for the following vpt_predict_batch_device calls on this workspace (default 0). */
vpt_status vpt_batch_set_flags(vpt_batch *b, unsigned flags);

/* Waits for the batch's last enqueued work and returns its device-side verdict. */
vpt_status vpt_batch_sync(vpt_batch *b);

/* Optional kernel timing: when enabled, vpt_predict_batch_device brackets the scoring kernel with HIP events
 * on the launch stream; after vpt_batch_sync, vpt_batch_kernel_ms returns that kernel's AVERAGE duration (ms)
 * over the calls made since the previous vpt_batch_kernel_ms (at most the 256 most recent) and the number of
 * workgroups (tiles) of the last call. */
/* How the last vpt_predict_batch_device call on this workspace cut its batch (diagnostics): the number of tiles, the flat
 * positions (chars + separators) a tile covers, and the kind -- 1 whole-sentence tiles, 2 tiles cut at any position with a halo
 * (batches with sentences too long for a tile), 0 the general kernels (models outside the packed shape). */
vpt_status vpt_batch_last_plan(const vpt_batch *b, uint32_t *n_tiles, uint32_t *tile_flat, uint32_t *kind);
vpt_status vpt_batch_set_timing(vpt_batch *b, int enabled);
vpt_status vpt_batch_kernel_ms(vpt_batch *b, float *score_kernel_ms, uint32_t *n_tiles);
/* The individual durations (ms, oldest first) of the timed calls since the previous vpt_batch_kernel_ms, at most `capacity`
 * of the 256 most recent; *n_out = how many were written (with ms_out == NULL: how many there are).  Does not reset. */
vpt_status vpt_batch_kernel_times(vpt_batch *b, float *ms_out, size_t capacity, size_t *n_out);

/* ---------------------------------------------------------------------------------------------------------
 * Sentence::fill_tags -> Predictor::predict_tags over a batch     (sentence.rs:1144-1148, predictor.rs:546-637)
 *
 * The predictor must have been created with predict_tags != 0 (the reference panics otherwise, predictor.rs:548-551:
 * here VPT_INVALID_ARGUMENT "this predictor is created with predict_tags = false").
 * n_tags  = the largest number of tag slots of any tag model (predictor.rs:466); 0 when the model has no tag models.
 * labels  : CharacterBoundary per boundary, laid out like labels_out of vpt_predict_batch -- normally its output,
 *           possibly edited by the caller's post-filters (as predict/src/main.rs:130-134 does between predict and
 *           fill_tags); VPT_BOUNDARY_UNKNOWN is honoured (tokens touching it get no tags).
 * tags_out: int32 [(total boundaries + n_sentences) * n_tags]: for char c (0-based) of sentence i, slot j:
 *           tags_out[(out_offsets[i] + i + c) * n_tags + j] = index of the chosen candidate in the matching tag
 *           model's j-th candidate list (model.rs:41-47), or -1 (None).  Only the LAST char of a token carries tags
 *           (predictor.rs:595-598), like Sentence::tags(). */
vpt_status vpt_predictor_n_tags(const vpt_predictor *p, uint32_t *n_tags);
vpt_status vpt_fill_tags_batch(const vpt_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets,
                               size_t n_sentences, const uint64_t *out_offsets, const uint8_t *labels,
                               int32_t *tags_out);
/* The same with VPT_FLAG_*: with VPT_FLAG_KYTEA_FULLWIDTH tokens and tag n-grams are matched on the normalised text,
 * as the CLI fills tags on the normalised sentence (predict/src/main.rs:156-170). */
vpt_status vpt_fill_tags_batch_flags(const vpt_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets,
                                     size_t n_sentences, const uint64_t *out_offsets, const uint8_t *labels,
                                     int32_t *tags_out, unsigned flags);

/* Device-resident variant of vpt_fill_tags_batch: all pointers are device pointers, asynchronous on `hip_stream`
 * (NULL = default stream).  d_labels as written by vpt_predict_batch_device (possibly edited by the caller's own
 * kernels).  The reference stores None for every char that does not end a token with a tag model (predictor.rs:558-573);
 * what this call leaves IN THE WORKSPACE is one record per token that HAS one (last char, tag model, chosen candidates),
 * sorted by position -- what vpt_write_tagged_batch_device reads.  d_tags_out: NULL (a tokenizer needs no more than the
 * records), or (total_boundaries + n_sentences) * n_tags int32 that receive the dense array described above (None = -1
 * everywhere else: a memset + a scatter of the records); vpt_expand_tags_batch_device makes it from the records later.
 * At most 2^32 - 257 chars per call.  The workspace keeps the decoded scalar values (4 bytes per char) between the
 * kernels; flags as set by vpt_batch_set_flags (fullwidth only).
 * When the call BEFORE this one on this workspace was vpt_predict_batch_device for the SAME buffers, sizes, flags and stream
 * (as Sentence::fill_tags follows Predictor::predict on the same sentence, predictor.rs:542), the chars that call decoded are
 * taken over and the decode kernel is skipped: do not rewrite d_utf8 in place between those two calls.  The chars are good
 * for that one call only, and only while no vpt_batch_sync lies between the two (a caller that waited for the device may have
 * rewritten its buffers): any other fill_tags call decodes the text it is given. */
vpt_status vpt_fill_tags_batch_device(const vpt_predictor *p, vpt_batch *b, const uint8_t *d_utf8,
                                      const uint64_t *d_byte_offsets, const uint64_t *d_out_offsets,
                                      size_t n_sentences, uint64_t total_boundaries, const uint8_t *d_labels,
                                      int32_t *d_tags_out, void *hip_stream);
/* Sentence::tags() of the batch of the last vpt_fill_tags_batch_device call on this workspace (same n_sentences and
 * total_boundaries): d_tags_out, (total_boundaries + n_sentences) * n_tags int32, receives the dense array -- None (-1)
 * for every char but the last one of a token that has a tag model (predictor.rs:556-598). */
vpt_status vpt_expand_tags_batch_device(const vpt_predictor *p, vpt_batch *b, size_t n_sentences, uint64_t total_boundaries,
                                        int32_t *d_tags_out, void *hip_stream);

/* ---------------------------------------------------------------------------------------------------------
 * Predictor::store_tag_scores(true) + Token::tag_candidates       (predictor.rs:510-514,599-601,632-634; sentence.rs:1218-1250)
 *
 * vpt_fill_tags_batch_flags, and besides the chosen candidates the i32 tag SCORES of every token that has a tag model --
 * the vector the reference keeps in sentence.tag_scores[i] when store_tag_scores is on: bias + the weights of the tag
 * n-grams that match around the token, laid out slot after slot over the slots with >= 2 candidates (a slot with one
 * candidate takes no entries; Token::tag_candidates reports score 0 for it).
 * vpt_predictor_tag_score_stride: the longest such vector over the predictor's tag models (0: none has scores).
 * tag_scores_out: int32 [(total boundaries + n_sentences) * stride] or NULL; row of char c of sentence i = out_offsets[i] + i + c;
 *                 the row of a token's LAST char holds its scores in entries [0, bias.len()); other rows read 0 (host variant)
 *                 or are left untouched (device variant).
 * tag_models_out: int32 [total boundaries + n_sentences] or NULL: index, in Model::tag_models order, of the tag model whose
 *                 token the token ending at that char is (the LAST one of a repeated token, predictor.rs:466-478); -1 where
 *                 no such token ends.  It says which rows of tag_scores_out are meaningful and whose candidate lists they score. */
vpt_status vpt_predictor_tag_score_stride(const vpt_predictor *p, uint32_t *stride);
vpt_status vpt_fill_tags_scores_batch(const vpt_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets,
                                      size_t n_sentences, const uint64_t *out_offsets, const uint8_t *labels, unsigned flags,
                                      int32_t *tags_out, int32_t *tag_scores_out, int32_t *tag_models_out);
vpt_status vpt_fill_tags_scores_batch_device(const vpt_predictor *p, vpt_batch *b, const uint8_t *d_utf8,
                                             const uint64_t *d_byte_offsets, const uint64_t *d_out_offsets,
                                             size_t n_sentences, uint64_t total_boundaries, const uint8_t *d_labels,
                                             int32_t *d_tags_out, int32_t *d_tag_scores_out, int32_t *d_tag_models_out,
                                             void *hip_stream);

/* ---------------------------------------------------------------------------------------------------------
 * Sentence::write_tokenized_text over a batch, boundary part                            (sentence.rs:850-886)
 *
 * Every sentence's tokens (the runs between WordBoundary labels, sentence.rs:1270-1300) joined by ' ', with a '\\' in
 * front of every ' ', '\\' and '/' of a surface -- what `vaporetto` prints for a model without tag models.  ("/tag"
 * suffixes are host-side strings: append them from vpt_fill_tags_batch's indices, as vaporetto_amd/api.py does.)
 * labels          : 0 / 1 per boundary, laid out like labels_out of vpt_predict_batch; VPT_BOUNDARY_UNKNOWN (only
 *                   partially annotated corpora have it, never predict) is VPT_INVALID_ARGUMENT.
 * text_out        : the tokenized sentences back to back, text_capacity bytes; 2 * (text bytes) + (chars) always
 *                   suffices.  Too small a capacity is VPT_INVALID_ARGUMENT, nothing useful is written.
 * text_offsets_out: [n_sentences + 1] byte range of every sentence's tokenized text in text_out. */
vpt_status vpt_write_tokenized_batch(const vpt_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets,
                                     size_t n_sentences, const uint64_t *out_offsets, const uint8_t *labels,
                                     uint8_t *text_out, uint64_t text_capacity, uint64_t *text_offsets_out);
/* The whole of write_tokenized_text for a predictor with tag models: Sentence::fill_tags on the given labels, then
 * every token followed by "/tag" for its slots up to the last Some -- an empty string for a None in between -- with
 * the same escaping (sentence.rs:866-881); what `vaporetto --predict-tags` prints.  flags: VPT_FLAG_KYTEA_FULLWIDTH
 * as for vpt_fill_tags_batch_flags (the tokens printed are the caller's text either way).  text_capacity: the bound
 * above + vpt_predictor_max_tag_suffix bytes per char.  A predictor without tag models writes what
 * vpt_write_tokenized_batch writes; one created with predict_tags == 0 is VPT_INVALID_ARGUMENT. */
vpt_status vpt_predictor_max_tag_suffix(const vpt_predictor *p, uint32_t *n_bytes);
vpt_status vpt_write_tagged_batch(const vpt_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets,
                                  size_t n_sentences, const uint64_t *out_offsets, const uint8_t *labels, unsigned flags,
                                  uint8_t *text_out, uint64_t text_capacity, uint64_t *text_offsets_out);
/* Predictor::predict (predictor.rs:518-543) AND Sentence::write_tokenized_text without tags (sentence.rs:850-886) for a batch in ONE scoring
 * launch: every tile of the specialised kernel writes the tokenized text of the chars it owns straight from its LDS, placed by a look-back
 * over the tiles' sizes (no second pass over the text, no writer launch).  d_scores / d_labels may be NULL (a tokenizer needs neither);
 * d_text_out needs 3 bytes per text byte at most, d_text_offsets_out [n_sentences + 1].  Asynchronous on hip_stream; errors (an output
 * larger than text_capacity among them) at vpt_batch_sync.  A model outside the specialised kernel's reach runs predict and the writer
 * one after the other. */
vpt_status vpt_predict_write_batch_device(const vpt_predictor *p, vpt_batch *b, const uint8_t *d_utf8, const uint64_t *d_byte_offsets,
                                          const uint64_t *d_out_offsets, size_t n_sentences, uint64_t total_boundaries,
                                          uint64_t max_sentence_bytes, int32_t *d_scores, uint8_t *d_labels, uint8_t *d_text_out,
                                          uint64_t text_capacity, uint64_t *d_text_offsets_out, void *hip_stream);

/* Device-resident variants: all pointers are device pointers, asynchronous on `hip_stream`; errors at vpt_batch_sync.
 * One kernel (kernels_emit.hip: a workgroup per run of sentences sizes, places and writes it).  The tagged one takes its
 * tags from the RECORDS a vpt_fill_tags_batch_device call left on the SAME workspace for the same batch AND THE SAME
 * LABELS (one per token that call found a tag model for: the model, the chosen candidates, the bytes they take; labels
 * changed in between are reported as offsets that do not match -- Sentence::fill_tags and write_tokenized_text see the
 * same boundaries too, sentence.rs:1144-1148, 850-886).  d_tags: not read (until round 6 the dense array of that
 * fill_tags call); NULL is fine. */
vpt_status vpt_write_tokenized_batch_device(const vpt_predictor *p, vpt_batch *b, const uint8_t *d_utf8,
                                            const uint64_t *d_byte_offsets, const uint64_t *d_out_offsets,
                                            size_t n_sentences, uint64_t total_boundaries, const uint8_t *d_labels,
                                            uint8_t *d_text_out, uint64_t text_capacity,
                                            uint64_t *d_text_offsets_out, void *hip_stream);
vpt_status vpt_write_tagged_batch_device(const vpt_predictor *p, vpt_batch *b, const uint8_t *d_utf8,
                                         const uint64_t *d_byte_offsets, const uint64_t *d_out_offsets,
                                         size_t n_sentences, uint64_t total_boundaries, const uint8_t *d_labels,
                                         const int32_t *d_tags, uint8_t *d_text_out, uint64_t text_capacity,
                                         uint64_t *d_text_offsets_out, void *hip_stream);

/* vpt_count_boundaries on the device (one wave per sentence + a prefix sum): d_out_offsets[n_sentences + 1]; the same
 * validation, reported at vpt_batch_sync. */
vpt_status vpt_count_boundaries_device(const vpt_predictor *p, vpt_batch *b, const uint8_t *d_utf8,
                                       const uint64_t *d_byte_offsets, size_t n_sentences, uint64_t *d_out_offsets,
                                       void *hip_stream);

/* Lines in, tokenized lines out -- the loop of predict/src/main.rs:122-176 for a whole batch: Sentence::from_raw,
 * [KyteaFullwidthFilter], Predictor::predict, [KyteaWsConstFilter / SplitLinebreaksFilter], [fill_tags],
 * write_tokenized_text.  Only the text crosses PCIe: char counting, scoring, tagging and the writer run on the
 * device.  flags: any VPT_FLAG_*; tagged != 0 needs a predictor created with predict_tags.  text_capacity:
 * 3 * (text bytes), plus (text bytes) * vpt_predictor_max_tag_suffix when tagged, always suffices. */
vpt_status vpt_tokenize_batch(const vpt_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets,
                              size_t n_sentences, unsigned flags, int tagged, uint8_t *text_out,
                              uint64_t text_capacity, uint64_t *text_offsets_out);

/* Diagnostics: when the environment variable VPT_PROFILE_PHASES is set at vpt_batch_create, the specialised
 * kernel accumulates, per workgroup (wave 0), the shader cycles spent in 0 text scan, 1 per-char decode,
 * 2 pattern lookups, 3 barrier wait, 4 boundary output.  Reads the sums (after a device sync) and resets them;
 * all zeros when profiling is off. */
vpt_status vpt_batch_phase_cycles(vpt_batch *b, uint64_t cycles[8]);
/* Diagnostics, same switch: the node reads the specialised kernel's lanes ISSUED since the last call, per level -- [0] unigram, [1] bigram,
 * [2] trigram nodes, [3] deep-trie entries, [4] deep-trie rows, [5] type rows in global memory (layout.h) -- i.e. the useful bytes of its
 * gathers (reads x node size) to set beside the 128-byte lines the L2 fetches for them; zeros without VPT_PROFILE_PHASES. */
vpt_status vpt_batch_node_reads(vpt_batch *b, uint64_t reads[8]);

/* ---------------------------------------------------------------------------------------------------------
 * Introspection (host-only; used by tests and the bench's roofline accounting)
 */
typedef struct vpt_model_info {
    uint32_t n_char_ngrams, n_type_ngrams, n_dict_words, n_tag_models;
    int32_t bias;
    uint32_t char_window, type_window;
    uint32_t max_pattern_chars;    /* longest char n-gram / dict word */
    uint32_t n_short_entries;      /* distinct strings of <= 3 chars (plus 3-char prefixes of longer ones) */
    uint32_t n_long_nodes;         /* trie nodes for strings of > 3 chars */
    uint32_t type_kind;            /* 0 none, 1 window table (cache variant), 2 pattern tables */
    uint64_t device_table_bytes;   /* bytes all tables occupy in HBM */
    uint64_t hot_table_bytes;      /* bytes of the tables the scoring kernel chosen for this model reads */
    uint32_t packed;               /* 1: the specialised kernel's packed tables (double-array trie, layout.h) are in use */
    uint32_t n_displaced;          /* hash-table keys that do not sit in their home slot */
    uint32_t type_rows;            /* type scores as rows added with a position's unigram row: 1 = the 294 rows in LDS (type n-grams of <= 3
                                      symbols), 2 = rows in global memory (up to 6 symbols); 0: window table / pattern tables */
    uint32_t n_overflow_children;  /* 0 (kept for layout compatibility: the double-array tables have no overflow) */
    uint32_t predict_tags;         /* the predict_tags flag the predictor was created with (0 from vpt_model_inspect without it) */
} vpt_model_info;

/* Model::read_slice's decoding (model.rs:127-135) without keeping the model: validates the bytes ("VaporettoTokenizer 0.5.0\n"
 * + bincode) and reports how many of them the model takes -- read_slice returns the rest to the caller.  Host only.
 * Errors: VPT_INVALID_MODEL "model version mismatch" | a decode error. */
vpt_status vpt_model_read_len(const uint8_t *model_bytes, size_t len, size_t *consumed);

/* Parses + validates + compiles the tables on the host only (no device).  Same errors as create. */
vpt_status vpt_model_inspect(const uint8_t *model_bytes, size_t len, int predict_tags, vpt_model_info *info);
vpt_status vpt_predictor_info(const vpt_predictor *p, vpt_model_info *info);

#ifdef __cplusplus
}
#endif
#endif /* VAPORETTO_HIP_H */
