"""include/vaporetto_hip.hpp -- the C++ mirror of the crate's Model / Predictor / Sentence on this path -- against the
Python mirror (vaporetto_amd/api.py) on the same library: a C++ driver (tests/native/cpp_mirror_test.cpp) is compiled and
run on the fixture model; tokens, scores, labels, char types, tokenized text with and without tags, the one-call
tokenizer and the error texts must be what the Python mirror (itself pinned by the reference's known answers) gives.
On the CPU the driver links the emulated build of the kernel sources (test infrastructure); `-m gpu` links the product."""
import os
import subprocess

import pytest

from tests import kat
from vaporetto_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "cpp_mirror_test.cpp")
LINES = ["まぁ社長は火星猫だ", "まぁ良いだろう", "火星猫", "あ", "12 ab/c\\d", "ＡＢＣ１２３ｱｲｳ漢字𠮷", "手\U0001f44f\U0001f3fdです\U0001f468\u200d\U0001f469か\u3099"]


def _build(lib_path: str, out: str) -> str:
    d, name = os.path.dirname(lib_path), os.path.basename(lib_path)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", out, SRC,
                           "-L" + d, "-l:" + name, "-Wl,-rpath," + d])
    return out


def _expected(tags: bool) -> str:
    raw, _ = kat.load_fixture("model.bin")
    pred = api.Predictor(api.Model.read_slice(raw)[0], tags)
    out = ["consumed %d of %d" % (len(raw), len(raw) + 1)]
    for l in LINES:
        s = api.Sentence.from_raw(l)
        pred.predict(s)
        out.append("tokens" + "".join(" [%s]" % t for t in s.iter_tokens()))
        out.append("scores" + "".join(" %d" % v for v in s.boundary_scores()))
        out.append("labels" + "".join(" %d" % v for v in s.boundaries()))
        out.append("types" + "".join(" %d" % v for v in s.char_types()))
        if tags:
            utf8, boff = api.pack_texts([l.encode("utf-8")])
            _, sc, md = pred.fill_tags_scores_packed(utf8, boff, api.count_boundaries(utf8, boff), s.boundaries())
            out.append("stored" + "".join(" %d:%d:%s" % (c, md[c], "".join("%d," % v for v in sc[c])) for c in range(len(md)) if md[c] >= 0))
            s.fill_tags()
        out.append("text " + s.write_tokenized_text())
        g = api.Sentence.from_raw(l)
        pred.predict(g)
        api.ConcatGraphemeClustersFilter().filter(g)
        out.append("graphemes " + g.write_tokenized_text())
    out += ["tokenize " + t for t in pred.tokenize(LINES, tagged=tags)]
    out.append("error 1 InvalidArgumentError: text: must contain at least one character -> [ ]")
    out.append("error 1 InvalidArgumentError: text: must not contain NULL")
    out.append("error 1 InvalidArgumentError: sentence: predict() has not been called")
    out.append("lifetime ok")
    out.append("error 0 model")
    return "\n".join(out) + "\n"


def _run(exe: str, tags: bool) -> str:
    model = os.path.join(ROOT, "tests", "golden", "model.bin")
    return subprocess.run([exe, model, "tags" if tags else "plain"], input="\n".join(LINES).encode("utf-8"), stdout=subprocess.PIPE,
                          check=True, timeout=600).stdout.decode("utf-8")


@pytest.mark.parametrize("tags", [False, True])
def test_cpp_mirror_on_the_emulated_sources(tags, tmp_path, monkeypatch):
    from tests import emu
    lib = emu.build_emulated()
    exe = _build(lib, str(tmp_path / "cpp_mirror_test"))
    monkeypatch.setattr(_lib, "_lib", emu.load())     # the Python mirror on the same (emulated) library
    assert _run(exe, tags) == _expected(tags)


@pytest.mark.gpu
@pytest.mark.parametrize("tags", [False, True])
def test_cpp_mirror_on_the_gpu(tags, tmp_path):
    exe = _build(_lib.LIB_PATH, str(tmp_path / "cpp_mirror_test"))
    assert _run(exe, tags) == _expected(tags)
