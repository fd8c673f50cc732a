"""
Common behaviour of the geometry objects: the numerical domain (``keep_inside=True``: everything outside is discarded) and
bodies cut out of it (``keep_inside=False``: everything inside is discarded).

API mirror of the reference's ``geometry/geometry_base.py`` (``GeometryObject``: verdict policy ``_apply_mask`` at
reference lines 40-76, common argument checks 78-107, abstract interface 109-222).  On the hot path the per-cell
predicate runs on the GPU: every built-in geometry describes itself through ``kernel_spec()`` and the refine loop hands
that description to the matching ``s3_mask_*`` kernel (include/s3hip.h).  ``check_cell`` is the host-side, single-cell
entry point with the reference's signature and truth table; a geometry without a ``kernel_spec`` is evaluated through it.
Constructor arguments are validated from each class's declarative ``_argument_rules()``.
"""
import logging
from abc import ABC, abstractmethod

from torch import Tensor

logger = logging.getLogger(__name__)


class GeometryObject(ABC):
    def __init__(self, name: str, keep_inside: bool, refine: bool = False, min_refinement_level: int = None):
        self._name, self._keep_inside = name, keep_inside
        self._refine, self._min_refinement_level = refine, min_refinement_level
        self._check_common_arguments()

    # -- verdict ---------------------------------------------------------------------------------------------------
    def _apply_mask(self, mask: Tensor, refine_geometry: bool) -> bool:
        """Cell verdict from the per-node "inside the geometry" mask:

        ===============  ===========================  ==========================
        mode             keep_inside=True (domain)    keep_inside=False (body)
        ===============  ===========================  ==========================
        remove cell?     no node inside               all nodes inside
        refine here?     not all nodes inside         any node inside
        ===============  ===========================  ==========================
        """
        inside, total = int(mask.sum()), mask.numel()
        if refine_geometry:
            return inside != total if self._keep_inside else inside > 0
        return inside == 0 if self._keep_inside else inside == total

    def _check_common_arguments(self) -> None:
        assert self._name != "", "A geometry object needs a non-empty name."
        assert isinstance(self._keep_inside, bool), f"keep_inside has to be a bool, got {type(self._keep_inside)}."
        if self._min_refinement_level is not None:
            if not self._refine:
                # a target level only makes sense for a refined geometry
                logger.warning(f"Geometry {self._name}: min_refinement_level={self._min_refinement_level} was given "
                               f"with refine=False; switching refine on.")
                self._refine = True
            assert self._min_refinement_level > 0, (f"min_refinement_level has to be positive, got "
                                                    f"{self._min_refinement_level}.")

    # -- read-only attributes --------------------------------------------------------------------------------------
    name = property(lambda self: self._name)
    keep_inside = property(lambda self: self._keep_inside)
    refine = property(lambda self: self._refine)
    min_refinement_level = property(lambda self: self._min_refinement_level)

    # -- interface of the concrete geometries --------------------------------------------------------------------------
    @abstractmethod
    def check_cell(self, cell_nodes: Tensor, refine_geometry: bool = False) -> bool:
        """verdict for one cell given its ``[2^d, d]`` node coordinates"""

    def kernel_spec(self):
        """``(kind, params...)`` consumed by the device mask kernels: ``("box", lo, hi)``, ``("sphere", pos, r)``,
        ``("cylinder", p0, axis, norm, r0, r1, is_cone)``, ``("polygon", xy[nv,2])``, ``("triangle", xy[3,2])``,
        ``("prism", origin, axis, norm, dims, xy[3,2])`` or ``("tetrahedra", pos[n,4,3], normals[n,3,4])``.

        ``None`` (the default): this geometry has no device predicate.  The refine loop then downloads the node coordinates
        of the cells in question and asks ``check_cell`` cell by cell -- any user-defined geometry with the reference's
        interface (s_cube.py:1816-1837 only ever calls ``check_cell(nodes, refine_geometry)``) keeps working, slowly."""
        return None

    # -- argument checks -------------------------------------------------------------------------------------------
    def _argument_rules(self) -> tuple:
        """the constructor-argument rules of the concrete geometry, declaratively: ``(holds, complaint)`` pairs, both
        callables without arguments (so that a later rule may rely on the earlier ones), checked in order"""
        return ()

    def _check_geometry(self) -> None:
        """raise ``AssertionError`` (the reference's error type for bad geometry arguments) at the first rule that fails"""
        for holds, complaint in self._argument_rules():
            assert holds(), f"Geometry '{self._name}': {complaint()}"

    def _validate(self) -> None:
        """what a constructor calls once its arguments are stored (the reference's classes call ``_check_geometry`` there, a
        name subclasses may still override)"""
        self._check_geometry()

    @property
    @abstractmethod
    def type(self) -> str:
        """short type tag (``"cube"``, ``"sphere"``, ``"cylinder"``, ``"coord_2D"``)"""

    @property
    @abstractmethod
    def main_width(self) -> float:
        """largest extent; the width of the root cell if this geometry is the domain"""

    @property
    @abstractmethod
    def center(self) -> Tensor:
        """centre; the centre of the root cell if this geometry is the domain"""

    @abstractmethod
    def _compute_main_width(self) -> float:
        pass

    @abstractmethod
    def _compute_center(self) -> Tensor:
        pass
