"""Driver-loop semantics of the reference's "Auto Train" (SURVEY §8f N4): UiFrame::update
(src/ui/UiFrame.cpp:266-298) without the GUI — budget throttle of AUTO_TRAIN_BUDGET = 100 steps/s
(src/Config.h:10), truth re-capture with randomly re-rotated camera spheres every `intervalCapture`
iterations (src/ui/tools/UiPanelToolsTruth.cpp:186-197), densify every `intervalDensify` iterations.

The truth renderer itself stays external (the reference uses OptiX): `capture(cameras) -> (framesW, framesB)`
is supplied by the caller."""
import random
import time

from . import camera as cam

AUTO_TRAIN_BUDGET = 100.0  # src/Config.h:10


class AutoTrainer:
    def __init__(self, trainer, project, capture, rng=None, clock=time.perf_counter):
        self.trainer, self.project, self.capture = trainer, project, capture
        self.rng = rng or random.Random(0)   # the reference uses C rand(); any uniform [0,1) source is equivalent
        self.clock = clock
        self.autoTraining = True
        self.autoTrainingBudget = 0.0
        self._last = clock()

    def randomRotate(self):
        """UiPanelToolsTruth::onButtonRandomRotate, src/ui/tools/UiPanelToolsTruth.cpp:192-197."""
        p = self.project
        p.sphere1.rotX = self.rng.random() * 360.0
        p.sphere1.rotY = self.rng.random() * 360.0
        p.sphere2.rotX = self.rng.random() * 360.0
        p.sphere2.rotY = self.rng.random() * 360.0

    def captureTruths(self):
        """UiPanelToolsTruth::onButtonCapture (:186-190) -> Trainer::captureTruths with the project's cameras."""
        cameras = cam.get_cameras_project(self.project)
        framesW, framesB = self.capture(cameras)
        self.trainer.captureTruths(cameras, framesW, framesB)

    def step(self):
        """One auto-train iteration: the body of `if(autoTrainingBudget >= 1.0f)` (src/ui/UiFrame.cpp:279-296)."""
        p = self.project
        capture = p.intervalCapture > 0 and p.iterations % p.intervalCapture == 0
        densify = p.intervalDensify > 0 and p.iterations % p.intervalDensify == 0
        if capture:
            self.randomRotate()
            self.captureTruths()
        self.trainer.train(p, densify)
        return capture, densify

    def update(self):
        """UiFrame::update (:266-298): called from the host's idle loop; returns True when an iteration ran."""
        now = self.clock()
        delta = now - self._last
        self._last = now
        self.project.previewTimer += delta
        if not self.autoTraining:
            return False
        self.autoTrainingBudget = min(1.0, self.autoTrainingBudget + delta * AUTO_TRAIN_BUDGET)
        if self.autoTrainingBudget >= 1.0:
            self.autoTrainingBudget = 0.0
            self.step()
            return True
        return False
