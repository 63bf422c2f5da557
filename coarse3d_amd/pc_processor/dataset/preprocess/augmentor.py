"""Point-cloud augmentation on the device (reference
pc_processor/dataset/preprocess/augmentor.py:7-230, same class names and parameters).

The random draws stay on the host in the reference's order (``random.uniform``, so a seeded run
draws the same flips / offsets / angles); the flip + translation + rotation they select are
applied to the CUDA point tensor by ONE kernel."""
import random

from scipy.spatial.transform import Rotation as R

from .... import ops


class AugmentParams(object):
    def __init__(self, p_flipx=0.0, p_flipy=0.0, p_transx=0.0, trans_xmin=0.0, trans_xmax=0.0, p_transy=0.0,
                 trans_ymin=0.0, trans_ymax=0.0, p_transz=0.0, trans_zmin=0.0, trans_zmax=0.0, p_rot_roll=0.0,
                 rot_rollmin=0.0, rot_rollmax=0.0, p_rot_pitch=0.0, rot_pitchmin=0, rot_pitchmax=0.0, p_rot_yaw=0.0,
                 rot_yawmin=0.0, rot_yawmax=0.0):
        self.setFlipProb(p_flipx, p_flipy)
        self.setTranslationParams(p_transx, trans_xmin, trans_xmax, p_transy, trans_ymin, trans_ymax, p_transz,
                                  trans_zmin, trans_zmax)
        self.setRotationParams(p_rot_roll, rot_rollmin, rot_rollmax, p_rot_pitch, rot_pitchmin, rot_pitchmax,
                               p_rot_yaw, rot_yawmin, rot_yawmax)

    def setFlipProb(self, p_flipx, p_flipy):
        self.p_flipx, self.p_flipy = p_flipx, p_flipy

    def setTranslationParams(self, p_transx=0.0, trans_xmin=0.0, trans_xmax=0.0, p_transy=0.0, trans_ymin=0.0,
                             trans_ymax=0.0, p_transz=0.0, trans_zmin=0.0, trans_zmax=0.0):
        self.p_transx, self.trans_xmin, self.trans_xmax = p_transx, trans_xmin, trans_xmax
        self.p_transy, self.trans_ymin, self.trans_ymax = p_transy, trans_ymin, trans_ymax
        self.p_transz, self.trans_zmin, self.trans_zmax = p_transz, trans_zmin, trans_zmax

    def setRotationParams(self, p_rot_roll=0.0, rot_rollmin=0.0, rot_rollmax=0.0, p_rot_pitch=0.0, rot_pitchmin=0,
                          rot_pitchmax=0.0, p_rot_yaw=0.0, rot_yawmin=0.0, rot_yawmax=0.0):
        self.p_rot_roll, self.rot_rollmin, self.rot_rollmax = p_rot_roll, rot_rollmin, rot_rollmax
        self.p_rot_pitch, self.rot_pitchmin, self.rot_pitchmax = p_rot_pitch, rot_pitchmin, rot_pitchmax
        self.p_rot_yaw, self.rot_yawmin, self.rot_yawmax = p_rot_yaw, rot_yawmin, rot_yawmax


class Augmentor(object):
    def __init__(self, params: AugmentParams):
        self.parmas = params          # (sic) the reference's attribute name

    def draw(self):
        """The reference's sequence of ``random.uniform`` draws (augmentor.py:182-228) ->
        (flip_x, flip_y, (tx, ty, tz), (roll, pitch, yaw) in degrees)."""
        p = self.parmas
        flip_x = random.uniform(0, 1) < p.p_flipx
        flip_y = random.uniform(0, 1) < p.p_flipy
        trans = []
        for prob, lo, hi in ((p.p_transx, p.trans_xmin, p.trans_xmax), (p.p_transy, p.trans_ymin, p.trans_ymax),
                             (p.p_transz, p.trans_zmin, p.trans_zmax)):
            trans.append(random.uniform(lo, hi) if random.uniform(0, 1) < prob else 0)
        rot = []
        for prob, lo, hi in ((p.p_rot_roll, p.rot_rollmin, p.rot_rollmax), (p.p_rot_pitch, p.rot_pitchmin, p.rot_pitchmax),
                             (p.p_rot_yaw, p.rot_yawmin, p.rot_yawmax)):
            rot.append(random.uniform(lo, hi) if random.uniform(0, 1) < prob else 0)
        return flip_x, flip_y, tuple(trans), tuple(rot)

    @staticmethod
    def apply(pointcloud, flip_x, flip_y, trans, rot_deg):
        """In place on a CUDA float32 [n, c>=3] tensor; rotation = R.from_euler('zyx', [yaw, pitch,
        roll], degrees=True) as in augmentor.py:167-174."""
        roll, pitch, yaw = rot_deg
        rot = R.from_euler("zyx", [yaw, pitch, roll], degrees=True).as_matrix()
        return ops.augment_points(pointcloud, -1.0 if flip_x else 1.0, -1.0 if flip_y else 1.0, trans, rot)

    def doAugmentation(self, pointcloud):
        return self.apply(pointcloud, *self.draw())
