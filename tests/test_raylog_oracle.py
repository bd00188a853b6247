"""The oracle's restatement of the reference ray log (raylogger.rs:54-66, pathtracer.rs:34-43,69-71) against the
oracle's own renderer: record counts equal the ray counts of the same render, the records of a path chain up
(the next extend ray starts within the surface offset of the logged hit point), shadow rays end on the light."""
import os

import numpy as np

import pathtracer_amd as pt
import pyoracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_record_layout_is_33_bytes():
    assert pt.RAYLOG_DTYPE.itemsize == 33
    assert [pt.RAYLOG_DTYPE.fields[k][1] for k in ("iteration", "px", "py", "bounces", "shadow", "intersect", "o", "d")] == [0, 2, 4, 6, 7, 8, 9, 21]


def test_raylog_counts_and_chaining():
    ref = orc.Scene(os.path.join(ROOT, "scenes", "cornell.obj"), max_bounces=5)
    w, h = 24, 16
    recs = ref.render_raylog(w, h, seed=2, first_sample=1, count=2).view(pt.RAYLOG_DTYPE)
    _, _, stats = ref.render(w, h, threads=1, iterations_per_thread=2, mode=1, seed=2, first_sample=1, stats=True)
    assert int((recs["shadow"] == 0).sum()) == stats["extend_rays"]
    assert int((recs["shadow"] == 1).sum()) == stats["shadow_rays"]
    # file order: iteration, row, column, bounce, extend before shadow
    key = (recs["iteration"].astype(np.int64) << 40) | (recs["py"].astype(np.int64) << 28) | (recs["px"].astype(np.int64) << 16) | (
        recs["bounces"].astype(np.int64) << 4) | recs["shadow"]
    assert (np.diff(key) >= 0).all()
    ext = recs[recs["shadow"] == 0]
    for a, b in zip(ext[:-1], ext[1:]):
        if (a["iteration"], a["px"], a["py"]) == (b["iteration"], b["px"], b["py"]) and b["bounces"] == a["bounces"] + 1:
            assert a["intersect"] == 1
            assert np.abs(a["o"] + a["d"] - b["o"]).max() < 1e-4  # hit point +- 1e-5 * n
    light = ref.export()["lights"][0]  # center, intensity, radius
    sh = recs[recs["shadow"] == 1]
    dist = np.linalg.norm(sh["o"] + sh["d"] - light[:3], axis=1)
    assert np.allclose(dist, light[6], rtol=1e-4)  # radius is the last of the 7
