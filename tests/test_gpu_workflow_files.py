"""File in, file out on the MI355X: gemma's own command lines (-g/-p/-c/-a or -bfile; -gk; -k ... -lmm; -eigen; -d/-u)
through tests/cpp/gemma_file_driver.cpp over the HIP library -- first-pass QC, kinship, the 10-digit hand-off, centring,
eigendecomposition, null model and the per-SNP loop all on the device, text parsed by the host thread pool -- against the
files the reference binary wrote for the same inputs (tests/golden/text/).  CPU twin: tests/test_file_driver_cpu.py.

Order: the first five (BXD, PLINK, -loco, multivariate PLINK, -lm) ran green on the MI355X in round 1's last GPU session; the
ones after them were written once that round's GPU minutes were spent -- same driver, same C ABI calls as
tests/test_gpu_reference.py makes through the Python mirror, host side verified by the CPU twin -- and come last so that a
surprise there cannot hide anything in front of them under `pytest -x`."""
import os

import pytest

import filecases as fc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    from gemma_amd import build
    build.build()
    return fc.build_driver(tmp_path_factory.mktemp("drv"), os.path.join(fc.ROOT, "gemma_amd"))


def test_bxd_bimbam_files_to_reference_outputs(driver, tmp_path):
    fc.bxd_bimbam_workflow(driver, tmp_path)


def test_plink_files_to_reference_outputs(driver, tmp_path):
    fc.plink_workflow(driver, tmp_path)


def test_loco_bimbam_files_to_reference_outputs(driver, tmp_path):
    fc.loco_workflow(driver, tmp_path, chrs=(2, 4), modes=(1, 4))


def test_mvlmm_plink_files_to_reference_outputs(driver, tmp_path):
    fc.mvlmm_workflow(driver, tmp_path, modes=(1, 2, 3, 4))


def test_lm_files_to_reference_outputs_and_golden_checksum(driver, tmp_path):
    fc.lm_workflow(driver, tmp_path)


def test_gxe_plink_files_to_reference_outputs(driver, tmp_path):
    fc.gxe_workflow(driver, tmp_path, modes=(1, 4))


def test_gene_expression_file_to_reference_outputs(driver, tmp_path):
    fc.gene_workflow(driver, tmp_path, modes=(1, 4))


def test_snps_notsnp_km2_to_reference_outputs(driver, tmp_path):
    fc.selection_options_workflow(driver, tmp_path)


def test_mvlmm_bimbam_text_to_reference_outputs(driver, tmp_path):
    fc.mvlmm_workflow(driver, tmp_path, modes=(1, 3), bimbam=True)


def test_hwe_filter_to_reference_outputs(driver, tmp_path):
    fc.hwe_workflow(driver, tmp_path)


def test_mvlmm_three_traits_missing_phenotypes(driver, tmp_path):
    fc.mvlmm3_workflow(driver, tmp_path, modes=(1, 3))


def test_standardised_kinship_from_text(driver, tmp_path):
    fc.standardised_kinship_workflow(driver, tmp_path)
