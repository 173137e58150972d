#!/usr/bin/env python3
"""Mine the numeric fixtures the reference's own tests hold for the hot path.

Runs ONLY in the build container (reads /root/reference as text; nothing of the
reference is imported, compiled or executed). It writes DATA (numbers) into
tests/golden/*.json; no reference source text is stored.

Fixtures extracted:
  * bspline_golden.json   - the eight 101-value Mathematica tables of
    trajectory_planning/splines/bspline_test.cc (kGoldenDataX .. Yppp) plus the
    inputs of BSplineGoldenTest.CompareToReference (:737-753) restated as data.
  * lp_regression.json    - the five literal 30-row LPs of
    trajectory_planning/time_optimal_path_timing_test.cc (:758-1072).
"""
import json
import os
import re
import sys

REF = "/root/reference/trajectory_planning"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)"


def numbers(text):
    return [float(x) for x in re.findall(NUM, text)]


def extract_quat_exp_tables():
    """QuatExpAndLogTest.ResultsMatchGoldenValues (splines/bsplineq_test.cc:99-198): 15 random
    quaternions (constructor order w, x, y, z) and Mathematica's Exp of them."""
    src = open(os.path.join(REF, "splines", "bsplineq_test.cc")).read()
    out = {}
    for key, name in (("input", "golden_input"), ("exp", "golden_exp_output")):
        m = re.search(r"%s\[kGoldenDataCount\]\s*=\s*\{(.*?)\};" % name, src, re.S)
        assert m, name
        vals = numbers(m.group(1))
        assert len(vals) == 60, (name, len(vals))
        out[key] = [vals[4 * i:4 * i + 4] for i in range(15)]
    fixture = {
        "source": "trajectory_planning/splines/bsplineq_test.cc:99-198 (Mathematica Exp[] of random quaternions)",
        "order": "w, x, y, z",
        "tolerance": "IsApprox: Eigen dummy_precision 1e-12 relative",
        "input": out["input"], "exp": out["exp"],
    }
    with open(os.path.join(OUT, "quat_exp_golden.json"), "w") as f:
        json.dump(fixture, f, indent=0)
    print("quat_exp_golden.json: 15 quaternion pairs")


def extract_bspline_tables():
    src = open(os.path.join(REF, "splines", "bspline_test.cc")).read()
    tables = {}
    for name in ["X", "Y", "Xp", "Yp", "Xpp", "Ypp", "Xppp", "Yppp"]:
        m = re.search(
            r"kGoldenData%s\[GetNumSamples\(\)\]\s*=\s*\{(.*?)\};" % name, src, re.S)
        assert m, name
        vals = numbers(m.group(1))
        assert len(vals) == 101, (name, len(vals))
        tables[name] = vals
    fixture = {
        "source": "trajectory_planning/splines/bspline_test.cc:97-726 (tables), :737-753 (inputs)",
        "degree": 3,
        "knots": [0, 0, 0, 0, 0.5, 1, 1, 1, 1],
        "control_points": [[1, 1], [2, 3], [3, -1], [4, 1], [5, 0]],
        "num_samples": 101,
        "tolerance_error_norm": 5e-14,
        "tables": tables,
    }
    with open(os.path.join(OUT, "bspline_golden.json"), "w") as f:
        json.dump(fixture, f)
    print("bspline_golden.json: 8 tables x 101 values")


def extract_lp_regression():
    src = open(os.path.join(REF, "time_optimal_path_timing_test.cc")).read()
    start = src.index("std::vector<LPInfo> lpinfo")
    end = src.index("for (", start)
    body = src[start:end]
    # each case: {30, {a...}, {b...}, {lower...}, {upper...}}
    cases = []
    pos = body.index("{") + 1
    # walk brace structure
    depth = 0
    cur = None
    groups = []
    i = pos
    while i < len(body):
        ch = body[i]
        if ch == "{":
            depth += 1
            if depth == 1:
                cur_start = i
                groups = []
            elif depth == 2:
                grp_start = i
        elif ch == "}":
            if depth == 2:
                groups.append(numbers(body[grp_start + 1:i]))
            elif depth == 1:
                head = body[cur_start + 1:body.index("{", cur_start + 1)]
                sz = int(numbers(head)[0])
                assert len(groups) == 4 and all(len(g) == sz for g in groups), (
                    sz, [len(g) for g in groups])
                cases.append({"size": sz, "a": groups[0], "b": groups[1],
                              "lower": groups[2], "upper": groups[3]})
            elif depth == 0:
                break
            depth -= 1
        i += 1
    assert len(cases) == 5, len(cases)
    fixture = {
        "source": "trajectory_planning/time_optimal_path_timing_test.cc:758-1072",
        "expected": "FindMaxSd2Simplex == FindMaxSd2BruteForce within 1e-8 (:1083-1086)",
        "cases": cases,
    }
    with open(os.path.join(OUT, "lp_regression.json"), "w") as f:
        json.dump(fixture, f)
    print("lp_regression.json: %d cases" % len(cases))


def extract_spline_utils_cases():
    """PolyLineToBspline3WaypointsPosesTest (splines/spline_utils_test.cc:31-146): the literal
    corners, radii and expected control poses of the corner-rounding tests. Poses are stored as
    (translation | rotation angle about the axis (1, 2, 3) / |(1, 2, 3)|), the two forms the tests
    use; numbers only."""
    src = open(os.path.join(REF, "splines", "spline_utils_test.cc")).read()
    cases = []
    for name in ("OneCorner", "Translation", "Rotation", "RadiusOutOfBounds", "ZeroRadius"):
        m = re.search(r"TEST\(PolyLineToBspline3WaypointsPosesTest, %s\)\s*\{(.*?)\n\}" % name, src, re.S)
        assert m, name
        body = m.group(1)
        corners = []
        for st in re.findall(r"corners\.emplace_back\((.*?)\);", body, re.S):
            a = re.search(r"AngleAxis<double>\(\s*(%s)," % NUM, st)
            v = re.search(r"Vector3d\((%s), (%s), (%s)\)" % (NUM, NUM, NUM), st)
            if a:
                corners.append({"translation": [0.0, 0.0, 0.0], "angle": float(a.group(1))})
            else:
                corners.append({"translation": [float(v.group(i)) for i in (1, 2, 3)], "angle": 0.0})
        radii = re.search(r"kTranslationalRadius = (%s);.*?kRotationalRadius = (%s);" % (NUM, NUM), body, re.S)
        if radii:
            tr, rr = float(radii.group(1)), float(radii.group(2))
        else:
            tr, rr = 0.0, 0.0          # OneCorner passes literal zeros
        expected = {}
        for mm in re.finditer(r"Pose3d cp_(\d)\((.*?)\);", body, re.S):
            a = re.search(r"AngleAxis<double>\(\s*(%s)," % NUM, mm.group(2))
            v = re.search(r"Vector3d\((%s), (%s), (%s)\)" % (NUM, NUM, NUM), mm.group(2))
            if a:
                expected[mm.group(1)] = {"translation": [0.0, 0.0, 0.0], "angle": float(a.group(1))}
            else:
                expected[mm.group(1)] = {"translation": [float(v.group(i)) for i in (1, 2, 3)], "angle": 0.0}
        count = int(re.search(r"ASSERT_EQ\((\d+), control_points\.size\(\)\)", body).group(1))
        cases.append({"name": name, "corners": corners, "translation_radius": tr, "rotation_radius": rr,
                      "num_control_points": count, "expected": expected})
    fixture = {"source": "trajectory_planning/splines/spline_utils_test.cc:31-146",
               "rotation_axis": [1.0, 2.0, 3.0],
               "rule": "control point 3i = corner i; 'expected' lists the other control points the tests pin "
                       "(ZeroRadius: 1 = corner 0, 2 = corner 1; OneCorner: all four = the corner)",
               "tolerance": "eigenmath IsApprox (1e-9 absolute used here)", "cases": cases}
    with open(os.path.join(OUT, "spline_utils_golden.json"), "w") as f:
        json.dump(fixture, f, indent=1)
    print("spline_utils_golden.json: %d cases" % len(cases))


if __name__ == "__main__":
    extract_quat_exp_tables()
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures are already committed")
    os.makedirs(OUT, exist_ok=True)
    extract_bspline_tables()
    extract_lp_regression()
    extract_spline_utils_cases()
