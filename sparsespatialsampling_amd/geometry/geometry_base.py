"""
Base class of the geometry objects that bound the numerical domain (``keep_inside=True``) or cut bodies out of it
(``keep_inside=False``).

API mirror of the reference's ``geometry/geometry_base.py`` (GeometryObject: ``_apply_mask`` :40-76, common argument
checks :78-107, abstract interface :109-222).  In this package the per-cell predicate on the hot path runs on the GPU:
every in-scope geometry describes itself through ``kernel_spec()`` and the refine loop hands that description to the
matching ``s3_mask_*`` kernel (include/s3hip.h).  ``check_cell`` remains as the host-side, single-cell entry point
with the reference's signature and truth table.
"""
import logging
from abc import ABC, abstractmethod

from torch import Tensor

logger = logging.getLogger(__name__)


class GeometryObject(ABC):
    def __init__(self, name: str, keep_inside: bool, refine: bool = False, min_refinement_level: int = None):
        self._name = name
        self._keep_inside = keep_inside
        self._refine = refine
        self._min_refinement_level = min_refinement_level
        self._check_common_arguments()

    def _apply_mask(self, mask: Tensor, refine_geometry: bool) -> bool:
        """Cell verdict from the per-node inside mask (truth table of reference geometry_base.py:40-76):

        ===============  ===========================  ==========================
        mode             keep_inside=True (domain)    keep_inside=False (body)
        ===============  ===========================  ==========================
        remove cells     no node inside               all nodes inside
        refine geometry  not all nodes inside         any node inside
        ===============  ===========================  ==========================
        """
        n_in, n = int(mask.sum()), mask.numel()
        if not refine_geometry:
            verdict = (n_in == 0) if self._keep_inside else (n_in == n)
        else:
            verdict = (n_in != n) if self._keep_inside else (n_in > 0)
        return bool(verdict)

    def _check_common_arguments(self) -> None:
        assert self._name != "", "Found empty string for the geometry object name. Please provide a name."
        assert isinstance(self._keep_inside, bool), (f"Invalid type for argument keep_inside. Expected bool but "
                                                     f"{type(self._keep_inside)} was given.")
        # a refinement level without refine=True means the user wants the geometry refined
        if not self._refine and self._min_refinement_level is not None:
            logger.warning(f"Found value refine={self._refine} while a min_refinement_level of "
                           f"{self._min_refinement_level} was provided for geometry {self._name}. Changing refine from"
                           f" {self._refine} to refine=True.")
            self._refine = True
        if self._refine and self._min_refinement_level is not None:
            assert self._min_refinement_level > 0, (f"Expected min_refinement_level > 0 but found "
                                                    f"min_refinement_level={self.min_refinement_level}.")

    @property
    def keep_inside(self):
        return self._keep_inside

    @property
    def name(self):
        return self._name

    @property
    def refine(self):
        return self._refine

    @property
    def min_refinement_level(self):
        return self._min_refinement_level

    @abstractmethod
    def check_cell(self, cell_nodes: Tensor, refine_geometry: bool = False) -> bool:
        pass

    @abstractmethod
    def kernel_spec(self) -> tuple:
        """``(kind, params...)`` consumed by the device mask kernels: ``("box", lo, hi)``, ``("sphere", pos, r)``,
        ``("cylinder", p0, axis, norm, r0, r1, is_cone)`` or ``("polygon", xy[nv,2])``."""

    @abstractmethod
    def _check_geometry(self) -> None:
        pass

    @property
    @abstractmethod
    def type(self) -> str:
        pass

    @property
    @abstractmethod
    def main_width(self) -> float:
        pass

    @property
    @abstractmethod
    def center(self) -> Tensor:
        pass

    @abstractmethod
    def _compute_main_width(self) -> float:
        pass

    @abstractmethod
    def _compute_center(self) -> Tensor:
        pass
