"""Python mirror of the reference's operator surface for this path.

* :class:`Args` / :func:`args_term`  <- ``Render_command.Args`` (render_command/src/render_command.ml:6-48)
* :class:`Integrator` (``create`` / ``render``) <- ``Integrator`` (path_tracer/src/integrator.mli:4-16)
* :func:`run` <- ``Render_command.Make(Scene).run`` (render_command.ml:64-109)

The opaque closures ``intersect`` / ``background`` of the reference's ``Scene`` argument are replaced by the
declarative scene (a :class:`path_tracer_ocaml_amd.host.HostScene` or any ``ptx_scene_desc``); everything runs
on the GPU through libptx_hip.so -- there is no CPU path here.
"""
import argparse
import time
from dataclasses import dataclass

import numpy as np

from . import Scene


@dataclass
class Args:  # Render_command.Args.t
    width: int
    height: int
    samples_per_pixel: int = 1
    output: str = "output.png"
    no_progress: bool = False
    max_bounces: int = 8


def args_term(parser=None):
    """Cmdliner term of the reference: -d/--dimension W,H (required), --samples-per-pixel (1), -o/--output
    (output.png), --no-progress, --max-ray-bounces (8)."""
    p = parser or argparse.ArgumentParser()

    def dimension(s):
        w, h = s.split(",")
        return int(w), int(h)

    p.add_argument("-d", "--dimension", type=dimension, required=True, metavar="WIDTH,HEIGHT", help="image dimensions")
    p.add_argument("--samples-per-pixel", type=int, default=1, metavar="INT", help="trace INT camera rays per pixel")
    p.add_argument("-o", "--output", default="output.png", metavar="PATH", help="write image to PATH")
    p.add_argument("--no-progress", action="store_true", help="suppress progress bar")
    p.add_argument("--max-ray-bounces", type=int, default=8, metavar="INT", help="max ray bounces")
    return p


def args_of_namespace(ns):
    w, h = ns.dimension
    return Args(w, h, ns.samples_per_pixel, ns.output, ns.no_progress, ns.max_ray_bounces)


class Integrator:
    """``Integrator.create ~width ~height ~image ~samples_per_pixel ~max_bounces ~camera ~intersect ~background``
    with (camera, intersect, background) folded into the declarative ``scene``; ``image`` is the (H, W, 3) f64
    array ``render`` fills, like the reference's Bimage."""

    def __init__(self, width, height, image, samples_per_pixel, max_bounces, scene, device=0):
        if image.shape != (height, width, 3) or image.dtype != np.float64:
            raise ValueError("image must be a float64 array of shape (height, width, 3)")
        self.width, self.height, self.image = width, height, image
        self.samples_per_pixel, self.max_bounces = samples_per_pixel, max_bounces
        self._scene = scene if isinstance(scene, Scene) else Scene(scene.ptr, device, keepalive=scene)
        self.stats = None

    @classmethod
    def create(cls, *, width, height, image, samples_per_pixel, max_bounces, scene, device=0):
        return cls(width, height, image, samples_per_pixel, max_bounces, scene, device)

    def render(self, update_progress=None):
        """``Integrator.render ~update_progress``: update_progress receives pixel counts summing to W*H."""
        rgb, st = self._scene.render(self.width, self.height, self.samples_per_pixel, self.max_bounces,
                                     progress=update_progress)
        self.image[...] = rgb
        self.stats = st
        return self.image


def run(args, scene, device=0, echo=print):
    """Render_command.Make(Scene).run: render, save the PNG, print ``rendered in: X ms``."""
    from . import host
    image = np.zeros((args.height, args.width, 3))
    integ = Integrator.create(width=args.width, height=args.height, image=image, samples_per_pixel=args.samples_per_pixel,
                              max_bounces=args.max_bounces, scene=scene, device=device)
    t0 = time.perf_counter()
    integ.render(None if args.no_progress else (lambda n: None))
    elapsed_ms = (time.perf_counter() - t0) * 1e3
    host.write_png(args.output, image)
    echo("rendered in: %.3f ms" % elapsed_ms)
    return image, integ.stats
