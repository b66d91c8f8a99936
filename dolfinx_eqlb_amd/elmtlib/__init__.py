"""Element library: hierarchic RT, Lagrange and quadrature tables without Basix."""

from .e_raviart_thomas import HierarchicRT, create_hierarchic_rt

__all__ = ["HierarchicRT", "create_hierarchic_rt"]
