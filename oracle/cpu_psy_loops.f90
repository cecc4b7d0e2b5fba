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

  ! ---- the GOcean `shallow` kernels (NE offset; formulas frozen in DESIGN.md section 6) as pointwise Fortran kernels and
  !      the seven PSy loop nests of one time step, each with the OpenMP `parallel do` over jj PSyclone's transformation
  !      emits: what a GOcean application runs on the CPU.  Same expression trees as the C oracle (checked bit for bit).
  pure subroutine compute_cu_code(i, j, cu, p, u)
    integer, intent(in) :: i, j
    real(c_double), intent(inout) :: cu(:, :)
    real(c_double), intent(in) :: p(:, :), u(:, :)
    cu(i, j) = 0.5_c_double * (p(i + 1, j) + p(i, j)) * u(i, j)
  end subroutine compute_cu_code
  pure subroutine compute_cv_code(i, j, cv, p, v)
    integer, intent(in) :: i, j
    real(c_double), intent(inout) :: cv(:, :)
    real(c_double), intent(in) :: p(:, :), v(:, :)
    cv(i, j) = 0.5_c_double * (p(i, j + 1) + p(i, j)) * v(i, j)
  end subroutine compute_cv_code
  pure subroutine compute_z_code(i, j, z, p, u, v, fsdx, fsdy)
    integer, intent(in) :: i, j
    real(c_double), intent(inout) :: z(:, :)
    real(c_double), intent(in) :: p(:, :), u(:, :), v(:, :), fsdx, fsdy
    z(i, j) = (fsdx * (v(i + 1, j) - v(i, j)) - fsdy * (u(i, j + 1) - u(i, j))) / &
              (p(i, j) + p(i + 1, j) + p(i + 1, j + 1) + p(i, j + 1))
  end subroutine compute_z_code
  pure subroutine compute_h_code(i, j, h, p, u, v)
    integer, intent(in) :: i, j
    real(c_double), intent(inout) :: h(:, :)
    real(c_double), intent(in) :: p(:, :), u(:, :), v(:, :)
    h(i, j) = p(i, j) + 0.25_c_double * (u(i, j) * u(i, j) + u(i - 1, j) * u(i - 1, j) + &
                                         v(i, j) * v(i, j) + v(i, j - 1) * v(i, j - 1))
  end subroutine compute_h_code
  pure subroutine compute_unew_code(i, j, unew, uold, z, cv, h, tdts8, tdtsdx)
    integer, intent(in) :: i, j
    real(c_double), intent(inout) :: unew(:, :)
    real(c_double), intent(in) :: uold(:, :), z(:, :), cv(:, :), h(:, :), tdts8, tdtsdx
    unew(i, j) = uold(i, j) + tdts8 * (z(i, j) + z(i, j - 1)) * &
                 (cv(i + 1, j) + cv(i, j) + cv(i, j - 1) + cv(i + 1, j - 1)) - tdtsdx * (h(i + 1, j) - h(i, j))
  end subroutine compute_unew_code
  pure subroutine compute_vnew_code(i, j, vnew, vold, z, cu, h, tdts8, tdtsdy)
    integer, intent(in) :: i, j
    real(c_double), intent(inout) :: vnew(:, :)
    real(c_double), intent(in) :: vold(:, :), z(:, :), cu(:, :), h(:, :), tdts8, tdtsdy
    vnew(i, j) = vold(i, j) - tdts8 * (z(i, j) + z(i - 1, j)) * &
                 (cu(i, j + 1) + cu(i - 1, j + 1) + cu(i - 1, j) + cu(i, j)) - tdtsdy * (h(i, j + 1) - h(i, j))
  end subroutine compute_vnew_code
  pure subroutine compute_pnew_code(i, j, pnew, pold, cu, cv, tdtsdx, tdtsdy)
    integer, intent(in) :: i, j
    real(c_double), intent(inout) :: pnew(:, :)
    real(c_double), intent(in) :: pold(:, :), cu(:, :), cv(:, :), tdtsdx, tdtsdy
    pnew(i, j) = pold(i, j) - tdtsdx * (cu(i, j) - cu(i - 1, j)) - tdtsdy * (cv(i, j) - cv(i, j - 1))
  end subroutine compute_pnew_code

  !> one time step = seven loop nests (intermediates over the box grown towards their consumers, as orc_sw_step)
  subroutine psy_shallow_step(prm, ld, ny, xs, xe, ys, ye, u, v, p, uold, vold, pold, cu, cv, z, h, unew, vnew, pnew, &
                              nthreads) bind(C, name="psy_shallow_step_f")
    real(c_double), intent(in) :: prm(5)          ! fsdx, fsdy, tdts8, tdtsdx, tdtsdy
    integer(c_int), value :: ld, ny, xs, xe, ys, ye, nthreads
    real(c_double), intent(in) :: u(ld, ny), v(ld, ny), p(ld, ny), uold(ld, ny), vold(ld, ny), pold(ld, ny)
    real(c_double), intent(inout) :: cu(ld, ny), cv(ld, ny), z(ld, ny), h(ld, ny), unew(ld, ny), vnew(ld, ny), pnew(ld, ny)
    integer :: i, j
    !$omp parallel num_threads(nthreads) private(i, j)
    !$omp do schedule(static)
    do j = ys, ye + 1
       do i = xs - 1, xe
          call compute_cu_code(i, j, cu, p, u)
       end do
    end do
    !$omp end do nowait
    !$omp do schedule(static)
    do j = ys - 1, ye
       do i = xs, xe + 1
          call compute_cv_code(i, j, cv, p, v)
       end do
    end do
    !$omp end do nowait
    !$omp do schedule(static)
    do j = ys - 1, ye
       do i = xs - 1, xe
          call compute_z_code(i, j, z, p, u, v, prm(1), prm(2))
       end do
    end do
    !$omp end do nowait
    !$omp do schedule(static)
    do j = ys, ye + 1
       do i = xs, xe + 1
          call compute_h_code(i, j, h, p, u, v)
       end do
    end do
    !$omp end do
    !$omp do schedule(static)
    do j = ys, ye
       do i = xs, xe
          call compute_unew_code(i, j, unew, uold, z, cv, h, prm(3), prm(4))
       end do
    end do
    !$omp end do nowait
    !$omp do schedule(static)
    do j = ys, ye
       do i = xs, xe
          call compute_vnew_code(i, j, vnew, vold, z, cu, h, prm(3), prm(5))
       end do
    end do
    !$omp end do nowait
    !$omp do schedule(static)
    do j = ys, ye
       do i = xs, xe
          call compute_pnew_code(i, j, pnew, pold, cu, cv, prm(4), prm(5))
       end do
    end do
    !$omp end do
    !$omp end parallel
  end subroutine psy_shallow_step

end module cpu_psy_loops
