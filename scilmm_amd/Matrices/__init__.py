"""Counterpart of the reference's ``scilmm/Matrices`` package for the pieces built here."""
from .Dominance import dominance  # noqa: F401
