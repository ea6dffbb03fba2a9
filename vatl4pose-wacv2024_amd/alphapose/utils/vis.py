"""Import-compatibility stub for the reference driver (scripts/Run_active_learning.py:43 imports
``vis_frame_fast`` / ``vis_frame`` at module level; they are only called with ``--vis``).

Drawing key-points on frames is host-side cv2 work outside the MI355X hot path:
the names exist so that the driver imports, calling them says so."""


def vis_frame_fast(*args, **kwargs):
    raise NotImplementedError("visualisation (cv2 drawing) is outside the MI355X hot path; run the reference's alphapose.utils.vis for --vis")


def vis_frame(*args, **kwargs):
    raise NotImplementedError("visualisation (cv2 drawing) is outside the MI355X hot path; run the reference's alphapose.utils.vis for --vis")
