"""Pins recorded from the reference: raw stdout of
`rnamotif -descr NAME gbrna.111.0.fastn` -- number of '>' lines and md5."""

SLACK = {
    "ire.descr": (0, "d41d8cd98f00b204e9800998ecf8427e"),
    "trna.descr": (1351, "1c8818d378175358e67ed039f508bed6"),
    "trna.efn.descr": (1351, "38aa3c197521f868e37361ae49d55473"),   # = descr/trna.descr (bits + efn)
    "pk1.descr": (193, "ebaeade0269a171de64abc76db9f0b06"),
    "qu+tr.descr": (9, "35451dce135e36651db71ea1d7138253"),
    "mp.ends.descr": (67, "27e9112f5d466a550a4a275a1bd11468"),
    "nanlin.descr": (13, "8004b7d6d662e03aa6f999eff274201b"),
    "pk_j1+2.descr": (32, "26c8cffcbcff82d90d5c38690151db8e"),
    "score.1.descr": (48, "f16b8258c11d13044cae5618b9c81c13"),
    "score.2.descr": (156, "f918789b405f46ef938463de9467f247"),
    "efn.descr": (445, "8e6d33e9b8f3b149bf14522e80f7079d"),
    "sprintf.descr": (139, "31ffe3ea736aad05f734b7757164178f"),
    "bulge.descr": (40, "8db885d1740f6e2589df8085221ad7d1"),
    "getbest.descr": (312, "3e3721a10d3396d35b1bef982101cffb"),
}

# `-sh -context -Dctx_maxlen=5 -descr NAME.strict.descr` (test/Makefile:139-245)
STRICT = {
    "nanlin": (0, "d41d8cd98f00b204e9800998ecf8427e"),
    "pk1": (45, "c8d40fa293a2c3a92b4b7d4d04370e40"),
    "pk_j1+2": (3, "5f8a252d1ba6fe8c4fddcbd0ceb054dd"),
    "qu+tr": (9, "b35ee52d83e358cb7c78d3b74294d720"),
    "score.1": (26, "4878662ffb9215575f925fdffd782c66"),
    "score.2": (86, "d9cb070279969646f50cefd5846a48f1"),
    "trna": (184, "a267b16ffdc3ab87863e4653975a0857"),
    "mp.ends": (46, "d840d1ce703e195e022823f592ae457a"),
    "efn": (327, "b9ba6b8004509a5b1c236ccd6b031ae8"),
    "sprintf": (82, "4ab6071653b867471d9a216dd5b0e5c3"),
    "bulge": (31, "620a36d5687d439f28e57ac763b622d8"),
    "getbest": (142, "e1917a54ce0f89d79fb35dc90b2d0aba"),
}
STRICT_ARGS = ["-sh", "-context", "-Dctx_maxlen=5"]

# syn10M (first 10 records of the synthetic database): hits per descriptor
SYN10M = {"trna.efn.descr": 630, "pk1.descr": 1039, "qu+tr.descr": 155, "mp.ends.descr": 36, "ire.descr": 3}
