"""Alias package so that code written against the reference's import paths
(``from openvoice import se_extractor``, ``from openvoice.api import ToneColorConverter`` -- the
upstream notebooks) runs on the MI355X implementation in ``openvoice_amd``."""
