//! Batch boundary scoring on an AMD MI355X behind the names of `vaporetto::Predictor`.
//!
//! `ffi` declares every entry point of `include/vaporetto_hip.h` (generated from the header by `bindings/rust/gen_ffi.py`;
//! `tests/test_rust_binding.py` in the MI355X repository compares the two, function by function, arity and widths).  This file
//! is the safe layer an application links against without touching the `vaporetto` crate: the model goes in as
//! `Model::to_vec()` bytes (model.rs:99-104), sentences come in as `vaporetto::Sentence`s, labels go back through
//! `Sentence::boundaries_mut()` (sentence.rs:1034) -- so `iter_tokens` / `write_tokenized_text` work on them as after
//! `Predictor::predict` (predictor.rs:518-543) -- and the i32 scores are returned beside them, because a `Sentence`'s score
//! buffer is `pub(crate)` (sentence.rs:92-94).  What needs crate-private access -- scores INSIDE the sentence, tags as
//! `Cow<'b, str>` borrowed from the model (predictor.rs:546-637; `TagModel::tags` is `pub(crate)`, model.rs:41-47) -- is the
//! in-crate variant of this shim, `vaporetto/src/hip.rs` behind a cargo feature, shown in INTEGRATION.md section 2.
//!
//! There is no CPU fallback: without a HIP device every call returns `HipError::Device`; keep the stock `Predictor` for that.
pub mod ffi;

use std::ffi::CStr;
use std::fmt;
use std::os::raw::c_int;

use vaporetto::{CharacterBoundary, Model, Sentence};

/// `VaporettoError`'s kinds (errors.rs:15-38) plus the device.
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum HipError {
    InvalidModel(String),
    InvalidArgument(String),
    Device(String),
}

impl fmt::Display for HipError {
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        match self {
            HipError::InvalidModel(m) | HipError::InvalidArgument(m) | HipError::Device(m) => f.write_str(m),
        }
    }
}
impl std::error::Error for HipError {}

pub type Result<T> = std::result::Result<T, HipError>;

fn check(st: c_int) -> Result<()> {
    if st == ffi::VPT_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::vpt_last_error()) }.to_string_lossy().into_owned();
    Err(match st {
        ffi::VPT_INVALID_MODEL => HipError::InvalidModel(msg),
        ffi::VPT_INVALID_ARGUMENT => HipError::InvalidArgument(msg),
        _ => HipError::Device(msg),
    })
}

/// `KyteaFullwidthFilter` before scoring, `KyteaWsConstFilter(t)` / `SplitLinebreaksFilter` on the labels (predict/src/main.rs:126-134).
/// `ConcatGraphemeClustersFilter` (`--wsconst G`, predict/src/main.rs:101-104) has no flag: it segments by UAX #29 and only edits labels, so
/// it stays on the host -- after `predict_batch` run `vaporetto_rules`' own filter over the sentences (their boundaries are the device's
/// labels by then), then `fill_tags` / `write_tokenized_text` as the CLI does; `tokenize_lines` cannot take it.
pub const FLAG_KYTEA_FULLWIDTH: u32 = 1;
pub const FLAG_SPLIT_LINEBREAKS: u32 = 1 << 7;
pub const fn flag_wsconst(char_type: u8) -> u32 {
    1 << char_type
}

/// `vaporetto::Predictor` for batches, on one GPU.
pub struct HipPredictor {
    raw: *mut ffi::vpt_predictor,
}
// immutable after creation; `&self` calls take a workspace of their own from the library's pool (the header's conventions) --
// what `Arc<Predictor>` is to vaporetto_tantivy/src/lib.rs:62-67
unsafe impl Send for HipPredictor {}
unsafe impl Sync for HipPredictor {}

impl HipPredictor {
    /// `Predictor::new(model, predict_tags)` (predictor.rs:450-508); the model by value, like the reference.
    pub fn new(model: Model, predict_tags: bool, device_id: i32) -> Result<Self> {
        let bytes = model.to_vec().map_err(|e| HipError::InvalidModel(e.to_string()))?;
        Self::from_model_bytes(&bytes, predict_tags, device_id)
    }

    /// The same from `Model::write`'s bytes ("VaporettoTokenizer 0.5.0\n" + bincode; zstd is outside the API, README.md:50-63).
    pub fn from_model_bytes(bytes: &[u8], predict_tags: bool, device_id: i32) -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::vpt_predictor_create(bytes.as_ptr(), bytes.len(), predict_tags as c_int, device_id, &mut raw) })?;
        Ok(Self { raw })
    }

    /// `Predictor::predict` over many sentences in one launch: every sentence's `boundaries_mut()` receives its labels; the
    /// i32 scores (`Sentence::boundary_scores()` in the reference, sentence.rs:1040-1046) come back per sentence.
    pub fn predict_batch(&self, sentences: &mut [Sentence<'_, '_>], flags: u32) -> Result<Vec<Vec<i32>>> {
        let (utf8, boff) = pack(sentences.iter().map(|s| s.as_raw_text()));
        let n = sentences.len();
        let mut ooff = vec![0u64; n + 1];
        check(unsafe { ffi::vpt_count_boundaries(utf8.as_ptr(), boff.as_ptr(), n, ooff.as_mut_ptr()) })?;
        let total = ooff[n] as usize;
        let (mut scores, mut labels) = (vec![0i32; total.max(1)], vec![0u8; total.max(1)]);
        check(unsafe {
            ffi::vpt_predict_batch_flags(self.raw, utf8.as_ptr(), boff.as_ptr(), n, scores.as_mut_ptr(), labels.as_mut_ptr(), ooff.as_ptr(), flags)
        })?;
        let mut out = Vec::with_capacity(n);
        for (i, s) in sentences.iter_mut().enumerate() {
            let (a, b) = (ooff[i] as usize, ooff[i + 1] as usize);
            for (dst, &l) in s.boundaries_mut().iter_mut().zip(&labels[a..b]) {
                *dst = if l == 1 { CharacterBoundary::WordBoundary } else { CharacterBoundary::NotWordBoundary };
            }
            out.push(scores[a..b].to_vec());
        }
        Ok(out)
    }

    /// The CLI's per-line loop for a batch of lines (predict/src/main.rs:122-176): `Sentence::from_raw`, the filters as flags,
    /// `predict`, `fill_tags` when `tagged`, `write_tokenized_text` -- only text crosses PCIe.  Returns one tokenized line per input.
    pub fn tokenize_lines<'a, I: IntoIterator<Item = &'a str>>(&self, lines: I, flags: u32, tagged: bool) -> Result<Vec<String>> {
        let (utf8, boff) = pack(lines.into_iter());
        let n = boff.len() - 1;
        let mut suffix = 0u32;
        if tagged {
            check(unsafe { ffi::vpt_predictor_max_tag_suffix(self.raw, &mut suffix) })?;
        }
        let cap = 3 * utf8.len() + 64 + utf8.len() * suffix as usize;
        let mut text = vec![0u8; cap];
        let mut toff = vec![0u64; n + 1];
        check(unsafe {
            ffi::vpt_tokenize_batch(self.raw, utf8.as_ptr(), boff.as_ptr(), n, flags, tagged as c_int, text.as_mut_ptr(), cap as u64, toff.as_mut_ptr())
        })?;
        Ok((0..n).map(|i| String::from_utf8_lossy(&text[toff[i] as usize..toff[i + 1] as usize]).into_owned()).collect())
    }

    /// `Predictor::serialize_to_vec` / `deserialize_from_slice_unchecked` (predictor.rs:640-664), in this library's own format
    /// (version and checksums checked): skips the table compile.
    pub fn serialize_to_vec(&self) -> Result<Vec<u8>> {
        let mut need = 0usize;
        check(unsafe { ffi::vpt_predictor_save(self.raw, std::ptr::null_mut(), 0, &mut need) })?;
        let mut buf = vec![0u8; need];
        check(unsafe { ffi::vpt_predictor_save(self.raw, buf.as_mut_ptr(), buf.len(), &mut need) })?;
        Ok(buf)
    }
    pub fn deserialize_from_slice(blob: &[u8], device_id: i32) -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::vpt_predictor_load(blob.as_ptr(), blob.len(), device_id, &mut raw) })?;
        Ok(Self { raw })
    }
    /// The same predictor on another GPU of the node: the tables are copied device to device, nothing is compiled again.
    pub fn clone_to_device(&self, device_id: i32) -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::vpt_predictor_clone_to_device(self.raw, device_id, &mut raw) })?;
        Ok(Self { raw })
    }
    pub fn info(&self) -> Result<ffi::vpt_model_info> {
        let mut info = ffi::vpt_model_info::default();
        check(unsafe { ffi::vpt_predictor_info(self.raw, &mut info) })?;
        Ok(info)
    }
}

impl Drop for HipPredictor {
    fn drop(&mut self) {
        unsafe { ffi::vpt_predictor_destroy(self.raw) }
    }
}

/// One batch over the GPUs of a node: `preds[r]` scores the r-th character-balanced range of the sentences (no data-path collective).
/// `boff` / `ooff`: the n + 1 byte / boundary offsets of the n sentences (`vpt_count_boundaries` makes the latter); `scores` and `labels` take
/// `ooff[n]` entries.  The C side trusts these sizes, so they are checked here: a safe function must not let a short slice become a write past it.
pub fn predict_batch_sharded(preds: &[&HipPredictor], utf8: &[u8], boff: &[u64], ooff: &[u64], scores: &mut [i32], labels: &mut [u8], flags: u32) -> Result<()> {
    let bad = |what: &str| Err(HipError::InvalidArgument(format!("InvalidArgumentError: {}", what)));
    if preds.is_empty() { return bad("preds: at least one predictor"); }
    if boff.is_empty() || ooff.len() != boff.len() { return bad("byte_offsets / out_offsets: n + 1 entries each"); }
    let n = boff.len() - 1;
    if boff.windows(2).any(|w| w[1] < w[0]) || ooff.windows(2).any(|w| w[1] < w[0]) { return bad("offsets: must not decrease"); }
    if (utf8.len() as u64) < boff[n] { return bad("utf8: shorter than byte_offsets[n]"); }
    if (scores.len() as u64) < ooff[n] || (labels.len() as u64) < ooff[n] { return bad("scores / labels: shorter than out_offsets[n]"); }
    let raws: Vec<*const ffi::vpt_predictor> = preds.iter().map(|p| p.raw as *const _).collect();
    check(unsafe {
        ffi::vpt_predict_batch_sharded(raws.as_ptr(), raws.len(), utf8.as_ptr(), boff.as_ptr(), n, scores.as_mut_ptr(), labels.as_mut_ptr(), ooff.as_ptr(), flags)
    })
}

fn pack<'a, I: Iterator<Item = &'a str>>(texts: I) -> (Vec<u8>, Vec<u64>) {
    let mut utf8 = Vec::new();
    let mut boff = vec![0u64];
    for t in texts {
        utf8.extend_from_slice(t.as_bytes());
        boff.push(utf8.len() as u64);
    }
    (utf8, boff)
}
