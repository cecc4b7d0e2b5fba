!> TEST INFRASTRUCTURE ONLY (the CPU side of bench.py's cpu_baseline leg and of tests/).
!!
!! The 5-point Jacobi step exactly as a GOcean/PSyclone application has it on the CPU: a pointwise
!! kernel with the dl_esm_inf calling convention (ji, jj, whole arrays; reference
!! infrastructure_mod.f90:30-41, kernel_mod.f90:28-50) and the PSy-layer loop nest over the
!! field's internal region that calls it once per point, optionally with the OpenMP
!! `parallel do` over jj that PSyclone's OpenMP transformation emits.  Built with
!! -ffp-contract=off so that it evaluates the same expression tree as the C oracle and the GPU.
module cpu_psy_loops
  use iso_c_binding
  implicit none
contains

  !> kern_code(ji, jj, out, in): one point of the update (SURVEY section 8, S5)
  pure subroutine jacobi5_code(ji, jj, fout, fin)
    integer, intent(in) :: ji, jj
    real(c_double), intent(inout) :: fout(:, :)
    real(c_double), intent(in) :: fin(:, :)
    fout(ji, jj) = 0.25_c_double * ((fin(ji - 1, jj) + fin(ji + 1, jj)) + (fin(ji, jj - 1) + fin(ji, jj + 1)))
  end subroutine jacobi5_code

  !> the PSy layer: do jj = ystart, ystop ; do ji = xstart, xstop ; call kern_code(...)
  subroutine psy_jacobi5(fin, fout, ld, ny, xstart, xstop, ystart, ystop, nthreads) bind(C, name="psy_jacobi5_f")
    integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop, nthreads
    real(c_double), intent(in) :: fin(ld, ny)
    real(c_double), intent(inout) :: fout(ld, ny)
    integer :: ji, jj
    if (nthreads > 1) then
       !$omp parallel do schedule(static) num_threads(nthreads) private(ji)
       do jj = ystart, ystop
          do ji = xstart, xstop
             call jacobi5_code(ji, jj, fout, fin)
          end do
       end do
       !$omp end parallel do
    else
       do jj = ystart, ystop
          do ji = xstart, xstop
             call jacobi5_code(ji, jj, fout, fin)
          end do
       end do
    end if
  end subroutine psy_jacobi5

end module cpu_psy_loops
