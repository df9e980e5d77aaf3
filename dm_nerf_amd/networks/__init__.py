"""Mirrors of the reference's ``networks`` package for the render hot path."""
