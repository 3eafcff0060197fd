function tables = nagp_ihgp_tables(A, Q, H, with_smoother)
% NAGP_IHGP_TABLES - the steady-state look-up tables of the infinite-horizon functions, flattened for nagp_mex
%
%   tables = nagp_ihgp_tables(A, Q, H, with_smoother)
%
% Per channel n (diagonal block ii of A, Q; row n of H) and per noise level ro_j, j = 1..32 on logspace(-2,4,32):
% the stationary predictive covariance PP = dare(A_ii', H_n,ii', Q_ii, ro_j), interpolated to the 200-point grid
% r = logspace(-2,4,200) with apxGrid('interp',...) (ihgp_ep_modulator_nmf.m:107-134); with_smoother also the
% steady-state smoother gain G and covariance PS2 (:150-191).  This is host set-up that stays MATLAB (Control System
% Toolbox `dare`); the kernels only look the rows up.
%
% Out: r (200 x 1), PP / PG (all channels, row-major per channel: n_grid rows of b^2 / 2 b^2 values),
%      pp_off / pg_off (int64, start of every channel in doubles)

  M = size(H,1); S = size(H,2);
  starts = [find(sum(H,1)) S+1];
  r = logspace(-2,4,200)';
  PP = []; PG = []; pp_off = zeros(M,1,'int64'); pg_off = zeros(M,1,'int64');
  for n = 1:M
    ii = starts(n):starts(n+1)-1;
    Ab = A(ii,ii); Qb = Q(ii,ii); hb = H(n,ii);
    ro = logspace(-2,4,32)';
    Pgrid = nan(numel(ro),numel(Qb));
    for j = 1:numel(ro)
      try
        P = dare(Ab',hb',Qb,ro(j));
        Pgrid(j,:) = P(:)';
      catch
        warning('nagp:dare','forward DARE %d of channel %d failed; grid point dropped',j,n);
        ro(j) = nan;
      end
    end
    Pgrid(isnan(ro),:) = []; ro(isnan(ro)) = [];
    U = apxGrid('interp',{ro},r,3);
    Pn = U*Pgrid;
    pp_off(n) = numel(PP);
    Pn = Pn'; PP = [PP; Pn(:)];                          %#ok<AGROW>  row-major: one grid row after the other
    if with_smoother
      Ggrid = nan(numel(ro),2*numel(Qb));
      for j = 1:numel(ro)
        P = reshape(Pgrid(j,:),size(Qb));
        K = P*hb'/(hb*P*hb'+ro(j));
        Pf = P - K*ro(j)*K';                             % as written in the reference (:162), not P - K*S*K'
        [Lc,notpd] = chol(Ab*Pf*Ab'+Qb,'lower');
        if notpd > 0
          error('nagp:tables','A*P*A''+Q of channel %d is not positive definite (the reference''s own branch for this case is broken, :167)',n);
        end
        G = Pf*Ab'/Lc'/Lc;
        QQ = Pf - G*P*G'; QQ = (QQ+QQ')/2;
        [V,E] = eig(QQ); keep = diag(E) > 0; QQ = V(:,keep)*E(keep,keep)*V(:,keep)';
        try
          PS2 = dare(G',0*G,QQ);
        catch
          PS2 = Pf*0; ro(j) = nan;
          warning('nagp:dare','smoother DARE %d of channel %d failed; grid point dropped',j,n);
        end
        Ggrid(j,:) = [PS2(:)' G(:)'];
      end
      Ggrid(isnan(ro),:) = []; ro(isnan(ro)) = [];
      U = apxGrid('interp',{ro},r,3);
      Gn = U*Ggrid;
      pg_off(n) = numel(PG);
      Gn = Gn'; PG = [PG; Gn(:)];                        %#ok<AGROW>
    else
      pg_off(n) = numel(PG);
      PG = [PG; zeros(2*numel(Qb)*numel(r),1)];          %#ok<AGROW>
    end
  end
  tables.r = r; tables.PP = PP; tables.PG = PG; tables.pp_off = pp_off; tables.pg_off = pg_off;
end
