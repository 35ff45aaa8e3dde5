"""The kernel SOURCES, run on the CPU: `tests/emu.py` compiles vaporetto_amd/csrc (the HIP kernels and the host side
of the C ABI, unchanged) with g++ against `tests/native/hipemu`, an emulator of the HIP execution model (fibers for
lanes; ballots, DPP, barriers, LDS, device allocations with red zones), and the GPU parity tests are then run against
that library.  What this pins without a GPU: the tile layout, the decode, the wave stacks and replays, the packed /
general / long-sentence paths, the tag kernels and the device-side error flags -- against the oracle, bit for bit.
What it cannot pin: anything about gfx950 itself (code generation, memory model, timing); `-m gpu` does that.

TEST INFRASTRUCTURE: the emulated library is loaded here and nowhere else; the product (`vaporetto_amd`) has no CPU
path and `tests/test_host_cabi.py::test_no_gpu_fails_loudly` keeps checking that."""
import os

import numpy as np
import pytest

from tests import devmem, emu
from tests import test_gpu_parity as G
from vaporetto_amd import _lib, api


@pytest.fixture(scope="module", autouse=True)
def _kernels_on_the_emulator():
    lib = emu.load()
    saved = _lib._lib
    _lib._lib = lib
    devmem.EMULATED = True
    yield
    import gc
    gc.collect()   # handles made by the emulated library are destroyed BY it (a predictor kept alive by a traceback until after the swap would be
    devmem.EMULATED = False   # handed to the real library's destructor, whose structs need not match a stale build's)
    _lib._lib = saved


def test_the_emulated_library_is_the_one_in_use():
    assert b"gfx950" in _lib.load().vpt_version()       # same sources, same version string
    assert os.path.basename(_lib.load()._name).startswith("libvaporetto_emu")   # (the default build, or the one VPT_EMU_DEFINES names)


# every GPU parity test that finishes in seconds on the emulator, as is
_AS_IS = [
    "test_boundary_kats", "test_predict_boundaries_like_reference", "test_fixture_splits", "test_appendix_scores",
    "test_sentence_reuse_and_overwrite", "test_random_models_vs_oracle", "test_window_sizes",
    "test_predict_tags_variant_same_scores", "test_ragged_and_edge_lengths", "test_ascii_and_four_byte_text",
    "test_packed_text_with_non_bmp_and_noncharacters", "test_batch_errors", "test_device_side_error_flags",
    "test_many_batches_through_one_predictor", "test_sentences_of_any_length_are_cut_across_tiles", "test_cut_tiles_with_tags_filters_unaligned_text_and_errors", "test_fast_and_general_kernels_agree_with_oracle",
    "test_packed_path_is_used_and_handles_wide_rows", "test_dense_packed_tables", "test_sparse_double_array_rows_interleave",
    "test_long_dictionary_words_cross_tile_sized_sentences", "test_very_long_words_and_compressed_chains",
    "test_non_bmp_pattern_models_use_the_general_tables", "test_deep_arena_limit_falls_to_the_general_kernels", "test_non_bmp_and_ffff_alphabets_on_the_packed_path", "test_long_type_ngrams_use_global_type_rows",
    "test_type_rows_and_window_table_agree_with_oracle", "test_type_weights_no_window_reads_and_padding_ngrams", "test_understated_length_bounds_are_reported",
    "test_tag_enabled_model_with_duplicate_type_ngrams_runs_on_the_packed_path", "test_predict_tags_like_reference",
    "test_fill_tags_requires_predict_tags_gpu", "test_fixture_tags_gpu", "test_random_tag_models_match_oracle",
    "test_tag_models_inside_and_outside_the_record_form", "test_fill_tags_front_end_stores_for_any_tag_count_and_alignment", "test_fill_tags_front_end_where_the_labels_run_out", "test_tag_models_under_wide_windows", "test_fill_tags_as_two_launches", "test_tag_front_end_over_runs_of_sentences", "test_tokenize_with_the_grapheme_cluster_filter", "test_stored_tag_scores_reproduce_the_scorer_kats", "test_stored_tag_scores_match_oracle_on_random_models", "test_tag_token_table_keys_and_queue",
    "test_converted_kytea_fixture_on_gpu", "test_fullwidth_filter_on_device", "test_label_post_filters_on_device",
    "test_device_resident_predict_then_fill_tags", "test_chars_left_by_predict_are_never_another_batchs", "test_fill_tags_with_offsets_that_do_not_match_the_text",
    "test_write_tokenized_text_on_device", "test_writer_every_byte_value_at_every_alignment", "test_writer_blocks_of_any_size", "test_concurrent_host_threads_share_a_predictor",
    "test_write_tagged_text_on_device", "test_predict_and_write_in_one_launch", "test_count_boundaries_on_the_device_flat_kernel", "test_tokenize_batch_into_pinned_buffers", "test_tokenize_batch_is_the_whole_pipeline",
    "test_tokenize_batch_in_chunks", "test_device_calls_accept_an_upper_bound_of_the_boundaries",
    "test_compiled_predictor_round_trip_and_clone", "test_compiled_predictor_rejects_damaged_blobs",
    "test_pipelined_host_path_matches_oracle", "test_labels_only_and_packed_tokenize_through_the_host_path",
    "test_sharded_predict_over_clones_equals_unsharded", "test_char_types_from_the_device",
]
for _name in _AS_IS:
    globals()[_name] = getattr(G, _name)
del _name


def test_lane_order_does_not_matter():
    """The scheduler visits the lanes of a workgroup in ascending order; in descending order (every cross-lane
    rendezvous is then reached by the HIGHEST lane first) a kernel that only orders its lanes through cross-lane
    operations must give the same results.  (The whole module also passes with HIPEMU_REVERSE=1.)"""
    lib = _lib.load()
    lib.hipemu_set_lane_order(1)
    try:
        for seed in (0, 7, 14):
            G.test_random_models_vs_oracle(seed)
        G.test_predict_tags_like_reference()
        G.test_tokenize_batch_is_the_whole_pipeline()
        G.test_device_side_error_flags()
    finally:
        lib.hipemu_set_lane_order(0)


def test_writer_long_tags_many_sentences_sized_down(monkeypatch):
    """test_writer_long_tags_many_sentences_and_long_sentences on 4 200 sentences (still more than one workgroup of the prefix sum)."""
    monkeypatch.setattr(G, "WRITER_TEST_SENTENCES", 4200)
    G.test_writer_long_tags_many_sentences_and_long_sentences()


def test_every_gpu_parity_test_is_accounted_for():
    """A new GPU parity test must be added to _AS_IS or to the sized-down list below."""
    sized_down = {"test_synthetic_configs_match_oracle", "test_batch_properties_at_full_config_size",
                  "test_writer_long_tags_many_sentences_and_long_sentences"}
    have = {n for n in dir(G) if n.startswith("test_") and callable(getattr(G, n))}
    assert have == set(_AS_IS) | sized_down, have ^ (set(_AS_IS) | sized_down)


@pytest.mark.parametrize("kind,scale,min_len,max_len,n", [(1, 0.02, 64, 64, 1500), (2, 0.02, 8, 512, 400), (2, 0.02, 1, 40, 2000)])
def test_synthetic_configs_sized_down(kind, scale, min_len, max_len, n):
    G.test_synthetic_configs_match_oracle(kind, scale, min_len, max_len, n)


def test_halves_and_permutation_sized_down():
    """test_batch_properties_at_full_config_size on 3 000 sentences."""
    from vaporetto_amd import synth
    raw = synth.synth_model(1, synth.SEED_BASE + 2, 0.02)
    n = 3000
    utf8, boff = synth.synth_sentences(raw, n, 64, 64, seed=synth.SEED_BASE + 2)
    pred = api.Predictor(api.Model.read_slice(raw)[0], False)
    scores, labels, _ = pred.predict_packed(utf8, boff)
    assert len(scores) == 63 * n and np.array_equal(labels, (scores > 0).astype(np.uint8))
    h = n // 2
    cut = int(boff[h])
    s1, _, _ = pred.predict_packed(utf8[:cut], boff[:h + 1])
    s2, _, _ = pred.predict_packed(utf8[cut:], boff[h:] - boff[h])
    assert np.array_equal(np.concatenate([s1, s2]), scores)
    perm = np.random.default_rng(5).permutation(n)
    parts = [utf8[int(boff[i]):int(boff[i + 1])] for i in perm]
    p_utf8 = np.concatenate(parts)
    p_boff = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    ps, _, _ = pred.predict_packed(p_utf8, p_boff)
    assert np.array_equal(ps.reshape(n, 63), scores.reshape(n, 63)[perm])


@pytest.mark.parametrize("env", [{}, {"VPT_TOKENIZE_CHUNK_BYTES": "700"}, {"VPT_EMU_DEFINES": "-DVPT_TAG_SUM_LOG2=7"}],
                         ids=["default", "small tokenize chunks", "a 128-bit summary of the token filter (a bit for many of the filter's)"])
def test_short_fuzz_of_the_kernel_sources(env):
    """tools/fuzz_gpu.py for a quarter of a minute on the emulator: random models (every window, tag models, wide weights) x ragged batches x
    flags against the oracle -- boundaries, tags, writers, vpt_tokenize_batch.  The seeds are fixed: a failure names the one to rerun."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, VPT_FUZZ_EMULATED="1", VPT_FUZZ_SEED0="400000", **env)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_gpu.py"), "12"], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode("utf-8", "replace")
    assert r.returncode == 0 and "no mismatch" in out, out[-2000:]


def test_writer_look_back_that_walks_to_the_front():
    """A build of the writer whose runs publish their SIZES only (-DVPT_EMIT_NO_PREFIX): every look-back walks back to the launch's front, and a
    run whose number is a multiple of 64 leaves the walk without having met the front's sentinel -- the case that, with chunks chained behind one
    another (vpt_tokenize_batch), lost the chunk's start on MI355X (round 6: the run's text landed in an earlier chunk's place; found by the GPU
    suite, one run in a few dozen).  One-sentence runs, chained chunks, poisoned output buffers -- in a process of its own (another library)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, VPT_EMU_DEFINES="-DVPT_EMIT_NO_PREFIX")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_kernel_emu.py"), "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "test_tokenize_batch_into_pinned_buffers or test_writer_blocks_of_any_size or test_tokenize_batch_in_chunks"],
                       env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200, cwd=root)
    out = r.stdout.decode("utf-8", "replace")
    assert r.returncode == 0 and " passed" in out, out[-3000:]
