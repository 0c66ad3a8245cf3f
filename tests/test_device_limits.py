"""What the device scanner refuses, and with which words (INTEGRATION.md lists the same): the
reference takes these descriptors; this build has no CPU search path to fall back to, so each is
refused before any scan -- at descriptor compilation or at rma_scanner_create(), whose host part
(rmd_build) runs before a device is asked for, so the refusals are testable here.  What is NOT
refused any more: up to 100 elements, the reference's own limit (compile.c:49); seq= expressions of
64 to 127 positions (round 3: a two-word automaton, tests/test_gpu_parity.py::test_long_seq_expressions); efn() / efn2()
back references and -- with iupac = 0 -- letters that are not acgt in seq= (round 4: the scan tests what the packed database
can tell, the host applies the expression at replay, tests/test_loose_seq.py); helices of 64 to 127 base pairs (round 4: a general instance with two-word sets of helix lengths, the reference's h3[ 101 ],
tests/test_gpu_parity.py::test_helices_of_64_to_127_base_pairs); calls over more than 15 helices (round 4: an instance of the energy kernel with stacks for the fifty helices a descriptor
can have, tests/test_gpu_parity.py::test_energy_calls_over_many_helices)."""
import os

import pytest

import rnamotif_amd as R

CASES = [
    ("helix of up to 130 base pairs",
     "descr\n\th5(minlen=4,maxlen=130)\n\t\tss(minlen=3,maxlen=8)\n\th3\n",
     "scanner", "helix element 1 allows 130 base pairs; the device scanner takes at most 127"),
    ("seq= that expands to more than 127 positions",
     'descr\n\tss(minlen=130,maxlen=150,seq="^' + "acgt" * 32 + '")\n',
     "scanner", "a seq= expression expands to more than 127 positions"),
    ("seq= of more than 128 atoms",
     'descr\n\tss(minlen=130,maxlen=150,seq="^' + "acgt" * 33 + '")\n',
     "compile", "cannot run on the device scanner: seq= pattern to"),
    ("back reference and mismatches in one seq=",
     'descr\n\tss(minlen=6,maxlen=6,seq="^\\(ac\\)g\\1",mismatch=1)\n',
     "compile", "mismatches"),
    ("more than 100 elements (the reference's own limit)",
     "descr\n" + "".join("\tss(len=1)\n" for _ in range(101)),
     "compile", "descr array size(100) exceeded."),
]


@pytest.mark.parametrize("what,text,where,message", CASES, ids=[c[0] for c in CASES])
def test_refusal_and_its_words(built, tmp_path, what, text, where, message):
    path = tmp_path / "x.descr"
    path.write_text(text)
    os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
    if where == "compile":
        with pytest.raises(R.RnamotifError) as e:
            R.Descriptor(["-descr", str(path)])
    else:
        d = R.Descriptor(["-descr", str(path)])
        with pytest.raises(R.RnamotifError) as e:
            R.Scanner(d)
    assert message in str(e.value), str(e.value)


def test_a_hundred_elements_are_taken(built, tmp_path):
    """48 and 100 elements, a helix of up to 100 base pairs and an efn() call over 17 helices pass the host part of
    rma_scanner_create(); without a GPU the only objection left is the missing device."""
    for text in ("descr\n" + "".join("\th5(minlen=2,maxlen=3)\n\t\tss(len=3)\n\th3\n\tss(len=1)\n" for _ in range(12)),
                 "descr\n" + "".join("\tss(len=1)\n" for _ in range(100)),
                 "descr\n\th5(minlen=4,maxlen=100)\n\t\tss(minlen=3,maxlen=8)\n\th3\n",
                 "descr\n" + "".join("\th5(tag='h%d',minlen=2,maxlen=3)\n\t\tss(len=3)\n\th3(tag='h%d')\n\tss(len=1)\n" % (i, i) for i in range(17)) +
                 "score\n\t{ SCORE = efn( h5['h0'], h3['h16'] ); }\n"):
        path = tmp_path / "y.descr"
        path.write_text(text)
        d = R.Descriptor(["-descr", str(path)])
        if R.lib().rma_device_count() == 0:
            with pytest.raises(R.RnamotifError, match="GPU only"):
                R.Scanner(d)
        else:
            R.Scanner(d).close()
